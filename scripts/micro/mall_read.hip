// Streaming-read bandwidth of one buffer re-read 64 times inside one launch, by buffer size: what the Infinity Cache (256 MiB)
// delivers to a plain float4 reader on this GPU, against the 8 TB/s the tiled step engine reaches.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) reader(const f4* __restrict__ p, size_t n4, float* out, int passes)
{
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int pass = 0; pass < passes; ++pass) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // 8 independent loads in flight per thread
    for (; i + 7 * stride < n4; i += 8 * stride) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) { const f4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    }
    if (s == 1.2345f) out[0] = s;
}

template <bool NT> void run(const f4* buf, size_t bytes, float* out, int blocks)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int reps = 64;                 // passes over the buffer inside ONE launch (as the step kernel's 64 timesteps)
    hipLaunchKernelGGL((reader<NT>), dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, out, 2);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((reader<NT>), dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, out, reps);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%6.0f MB  %s  blocks %5d: %.2f TB/s\n", bytes / 1e6, NT ? "nt     " : "default", blocks, (double)bytes * reps / (ms * 1e-3) / 1e12);
}

int main()
{
    const size_t max_bytes = (size_t)1 << 30;
    f4* buf; float* out;
    (void)hipMalloc(&buf, max_bytes); (void)hipMalloc(&out, 64);
    (void)hipMemset(buf, 0, max_bytes);
    for (size_t mb : {28, 56, 112, 168, 223, 446, 1024})
        for (int blocks : {1024, 2048, 4096}) {
            run<false>(buf, mb * 1000000 / 4096 * 4096, out, blocks);
            run<true>(buf, mb * 1000000 / 4096 * 4096, out, blocks);
        }
    return 0;
}
