// tools/ubench/empty_launch.hip -- what an early-exit kernel costs on the stream, by grid size (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o empty_launch empty_launch.hip ; run on the GPU box.
// A "work" kernel of ~60 us, then N kernels whose every workgroup loads one flag and returns (K1's verify and chained
// launches and an unused fallback look like that), then another work kernel; wall time per idle kernel from HIP events
// around the whole chain, minus the chain without idle kernels.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(512) void k_idle(const unsigned* flag, unsigned* out)
{
    __shared__ unsigned lds[11000];   // 44 KB, as K1/K2 ask for
    if (*flag == 0) return;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[blockIdx.x] = lds[(threadIdx.x + 1) & 511];
}

__global__ __launch_bounds__(512) void k_work(unsigned* out, int iters)
{
    unsigned v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;
    if (v == 12345u) out[blockIdx.x] = v;
}

int main()
{
    unsigned *flag, *out;
    hipMalloc(&flag, 4);
    hipMalloc(&out, 1 << 20);
    hipMemset(flag, 0, 4);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grids[] = {0, 1, 64, 256, 714, 2857};
    for (int threads : {64, 512})
        for (int g : grids) {
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                hipEventRecord(e0, s);
                for (int q = 0; q < 10; ++q) {
                    hipLaunchKernelGGL(k_work, dim3(714), dim3(512), 0, s, out, 20000);
                    if (g)
                        for (int n = 0; n < 3; ++n) hipLaunchKernelGGL(k_idle, dim3(g), dim3(threads), 0, s, flag, out);
                }
                hipEventRecord(e1, s);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("idle kernels of %4d x %3d threads: chain of 10 x (work + 3 idle) = %.1f us\n", g, threads, best * 1e3f);
        }
    return 0;
}
