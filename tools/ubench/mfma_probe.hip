// tools/ubench/mfma_probe.hip -- what v_mfma_f32_4x4x1_16b_f32 does on gfx950 and what it costs beside VALU work.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run on the GPU box.
//   1. operand layout: which lane/register holds A_b[i], B_b[j], D_b[i][j] of the 16 independent 4x4 blocks;
//   2. arithmetic: a chain of them accumulates exactly as fmaf() does (bitwise);
//   3. issue: wave-cycles per loop iteration of VALU work alone, MFMAs alone and both together, at W waves per SIMD
//      (does the matrix pipe run beside the VALU, and what does issuing an MFMA take from the VALU's slots).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void k_layout(float* out)
{
    const int lane = threadIdx.x;
    const float a = 1.0f + (float)lane / 128.0f;     // mantissa names the A lane
    const float b = ldexpf(1.0f, lane);              // exponent names the B lane
    float4v c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

// chain: D = sum_k A_k * B_k accumulated by successive MFMAs (k = 0..7) against fmaf in the same order
__global__ void k_chain(const float* av, const float* bv, float* out)
{
    const int lane = threadIdx.x;
    float4v c = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 8; ++k) c = __builtin_amdgcn_mfma_f32_4x4x1f32(av[k * 64 + lane], bv[k * 64 + lane], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

#define REP4(X) X X X X
#define REP8(X) X X X X X X X X
template <int WHICH>
__global__ __launch_bounds__(256) void k_rate(int iters, float* out, unsigned long long* cyc)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c0 = 1.0001f;
    float4v m0 = {a0, a1, a2, a3}, m1 = m0, m2 = m0, m3 = m0;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (WHICH == 0)   // 32 VALU (fast class)
            asm volatile(REP8("v_fma_f32 %0, %0, %4, %1\n\tv_fma_f32 %1, %1, %4, %2\n\tv_fma_f32 %2, %2, %4, %3\n\tv_fma_f32 %3, %3, %4, %0\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0));
        if (WHICH == 1)   // 8 MFMA, four independent accumulators
            asm volatile(REP4("v_mfma_f32_4x4x1_16b_f32 %0, %4, %5, %0\n\tv_mfma_f32_4x4x1_16b_f32 %1, %4, %5, %1\n\t")
                         REP4("v_mfma_f32_4x4x1_16b_f32 %2, %4, %5, %2\n\tv_mfma_f32_4x4x1_16b_f32 %3, %4, %5, %3\n\t")
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(c0), "v"(a0));
        if (WHICH == 2)   // 32 VALU with 8 MFMA spread between them
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %1\n\tv_fma_f32 %1, %1, %8, %2\n\tv_mfma_f32_4x4x1_16b_f32 %4, %8, %9, %4\n\tv_fma_f32 %2, %2, %8, %3\n\tv_fma_f32 %3, %3, %8, %0\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(c0), "v"(c0));
        if (WHICH == 3)   // 32 VALU, then 8 MFMA in a block (independent accumulators two by two)
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %1\n\tv_fma_f32 %1, %1, %8, %2\n\tv_fma_f32 %2, %2, %8, %3\n\tv_fma_f32 %3, %3, %8, %0\n\t")
                         REP4("v_mfma_f32_4x4x1_16b_f32 %4, %8, %9, %4\n\tv_mfma_f32_4x4x1_16b_f32 %5, %8, %9, %5\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(c0), "v"(c0));
        if (WHICH == 4)   // 32 DPP-class VALU (2.7-cycle class)
            asm volatile(REP8("v_fmac_f32_dpp %0, %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %2, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f32_dpp %2, %3, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %0, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0));
        if (WHICH == 5)   // 32 DPP-class VALU with 8 MFMA spread between them
            asm volatile(REP8("v_fmac_f32_dpp %0, %1, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %2, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                              "v_mfma_f32_4x4x1_16b_f32 %4, %8, %9, %4\n\t"
                              "v_fmac_f32_dpp %2, %3, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %0, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(c0), "v"(c0));
        if (WHICH == 6)   // 8 MFMA, ONE accumulator (a dependent chain, as the k-steps of one product are)
            asm volatile(REP8("v_mfma_f32_4x4x1_16b_f32 %0, %4, %5, %0\n\t")
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(c0), "v"(a0));
        if (WHICH == 7)   // 32 VALU + 16 MFMA spread
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %1\n\tv_mfma_f32_4x4x1_16b_f32 %4, %8, %9, %4\n\tv_fma_f32 %1, %1, %8, %2\n\tv_fma_f32 %2, %2, %8, %3\n\tv_mfma_f32_4x4x1_16b_f32 %5, %8, %9, %5\n\tv_fma_f32 %3, %3, %8, %0\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(c0), "v"(c0));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + m0[0] + m1[1] + m2[2] + m3[3];
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = t1 - t0;
        cyc[gridDim.x + blockIdx.x] = r1 - r0;
    }
}

template <int WHICH>
void rate(const char* name, int W)
{
    const int iters = 2000, ncu = 256;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)ncu * W * 256 * 4);
    hipMalloc(&cyc, (size_t)ncu * W * 8 * 2);
    hipLaunchKernelGGL(k_rate<WHICH>, dim3(ncu * W), dim3(256), 0, 0, 10, out, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_rate<WHICH>, dim3(ncu * W), dim3(256), 0, 0, iters, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(ncu * W * 2);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0, real = 0;
    const size_t nb = h.size() / 2;
    for (size_t i = 0; i < nb; ++i) avg += (double)h[i], real += (double)h[nb + i];
    avg /= nb, real /= nb;
    printf("%-52s W=%d  wave-cycles/iteration %8.1f  SIMD-cycles/iteration %7.1f  clock %.0f MHz\n", name, W, avg / iters, avg / iters / W, avg / real * 100.0);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    float* d;
    hipMalloc(&d, 64 * 4 * 4);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    printf("layout of v_mfma_f32_4x4x1_16b_f32: D lane L register r = A(lane la) * B(lane lb)\n");
    bool as_expected = true;
    for (int L = 0; L < 64; ++L) {
        printf("  lane %2d:", L);
        for (int r = 0; r < 4; ++r) {
            int e;
            const float m = frexpf(h[L * 4 + r], &e);    // value = m * 2^e, m in [0.5, 1)
            const int lb = e - 1, la = (int)lrintf((m * 2.0f - 1.0f) * 128.0f);
            printf("  r%d = A[%2d]*B[%2d]", r, la, lb);
            if (la != (L & ~3) + r || lb != L) as_expected = false;
        }
        printf("\n");
    }
    printf("layout as expected (D[lane 4b+j][reg i] = A[lane 4b+i] * B[lane 4b+j]): %s\n", as_expected ? "yes" : "NO");

    // chain against fmaf
    std::vector<float> av(512), bv(512), want(256, 0.f);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) % 200001 - 100000) / 977.0f; };
    for (auto& x : av) x = rnd();
    for (auto& x : bv) x = rnd() * 0.01f;
    for (int L = 0; L < 64; ++L)
        for (int r = 0; r < 4; ++r) {
            float acc = 0.f;
            for (int k = 0; k < 8; ++k) acc = fmaf(av[k * 64 + (L & ~3) + r], bv[k * 64 + L], acc);
            want[L * 4 + r] = acc;
        }
    float *da, *db;
    hipMalloc(&da, 2048), hipMalloc(&db, 2048);
    hipMemcpy(da, av.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(db, bv.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, da, db, d);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += memcmp(&h[i], &want[i], 4) != 0;
    printf("8-step MFMA chain against fmaf in the same order: %d of 256 results differ\n", bad);

    for (int W : {1, 4}) {
        rate<0>("32 v_fma_f32", W);
        rate<1>("8 mfma 4x4x1 (4 accumulators)", W);
        rate<6>("8 mfma 4x4x1 (1 accumulator, dependent)", W);
        rate<2>("32 v_fma_f32 + 8 mfma spread", W);
        rate<3>("32 v_fma_f32, then 8 mfma", W);
        rate<7>("32 v_fma_f32 + 16 mfma spread", W);
        rate<4>("32 v_fmac_f32_dpp", W);
        rate<5>("32 v_fmac_f32_dpp + 8 mfma spread", W);
    }
    return 0;
}
