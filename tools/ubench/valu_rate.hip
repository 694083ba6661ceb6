// tools/ubench/valu_rate.hip -- issue cost of the VALU instructions K4 is built from (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
// For each instruction: ITER x 64 independent instances per wave; W waves per SIMD.
// Prints cycles per instruction per SIMD = elapsed_cycles * (#SIMD-resident waves share) / count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define BODY(INS)                                                                              \
    for (int it = 0; it < iters; ++it) {                                                       \
        asm volatile(REP8(INS "\n\t") REP8(INS "\n\t") REP8(INS "\n\t") REP8(INS "\n\t")          \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) \
                     : "v"(c0), "v"(c1));                                                     \
    }

typedef float float2v __attribute__((ext_vector_type(2)));

template <int WHICH>
__global__ __launch_bounds__(256) void k(int iters, float* out, unsigned long long* cyc)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float2v b0 = {a0, a1}, b1 = {a1, a2}, b2 = {a2, a3}, b3 = {a3, a0};
    float c0 = 1.0001f;
    float2v c1 = {0.9999f, 1.0001f};
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (WHICH == 0) { BODY("v_fma_f32 %0, %0, %8, %1") }
    if (WHICH == 1) { BODY("v_pk_fma_f32 %4, %4, %9, %5") }
    if (WHICH == 2) { BODY("v_pk_mul_f32 %4, %4, %9") }
    if (WHICH == 3) { BODY("v_pk_add_f32 %4, %4, %9") }
    if (WHICH == 4) { BODY("v_mul_f32_dpp %0, %1, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf") }
    if (WHICH == 5) { BODY("v_fmac_f32_dpp %0, %1, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf") }
    if (WHICH == 6) { BODY("v_rndne_f32 %0, %1") }
    if (WHICH == 7) { BODY("v_floor_f32 %0, %1") }
    if (WHICH == 8) { BODY("v_cvt_pk_u8_f32 %0, %1, 1, %0") }
    if (WHICH == 9) { BODY("v_cvt_f32_i32 %0, %1") }
    if (WHICH == 10) { BODY("v_max_f32 %0, %0, %1") }
    if (WHICH == 11) { BODY("v_add_f32 %0, %0, %8") }
    if (WHICH == 12) { BODY("v_mul_f32 %0, %0, %8") }
    if (WHICH == 13) { BODY("v_fmac_f32 %0, %1, %8") }
    if (WHICH == 14) { BODY("v_max3_f32 %0, %0, %1, %2") }
    if (WHICH == 15) { BODY("v_fmac_f32_dpp %0, %0, %8 row_half_mirror row_mask:0xf bank_mask:0xf") }
    if (WHICH == 16) { BODY("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD src0_sel:WORD_1") }
    if (WHICH == 17) { BODY("v_cndmask_b32 %0, %1, %2, vcc") }
    if (WHICH == 18) { BODY("v_cmp_eq_f32 vcc, %0, %1") }
    if (WHICH == 42) { BODY("v_cndmask_b32_e64 %0, %1, %2, s[10:11]") }
    if (WHICH == 43) { BODY("v_cndmask_b32 %0, %1, %2, vcc\n\tv_cndmask_b32 %1, %2, %3, vcc\n\tv_cndmask_b32 %2, %3, %0, vcc\n\tv_cndmask_b32 %3, %0, %1, vcc") }
    if (WHICH == 44) { BODY("v_bfi_b32 %0, %1, %2, %3") }
    if (WHICH == 45) { BODY("v_cndmask_b32_e64 %0, 0, 1, s[10:11]") }
    if (WHICH == 46) { BODY("v_ashrrev_i32 %0, 31, %1") }
    if (WHICH == 47) { BODY("v_min_u32 %0, %0, %1") }
    if (WHICH == 48) { BODY("v_cndmask_b32_e64 %0, %1, %2, s[10:11]\n\tv_add_u32 %1, %1, %3") }
    if (WHICH == 19) { BODY("v_and_b32 %0, %0, %1") }
    if (WHICH == 20) { BODY("v_lshlrev_b32 %0, 1, %1") }
    if (WHICH == 21) { BODY("v_add_u32 %0, %0, %1") }
    if (WHICH == 22) { BODY("v_sub_f32 %0, %0, %8") }
    if (WHICH == 23) { BODY("v_add_f32_e64 %0, %8, |%1|") }
    if (WHICH == 24) { BODY("v_fma_f32 %0, |%0|, %8, %1") }
    if (WHICH == 25) { BODY("v_min_f32 %0, %0, %1") }
    if (WHICH == 26) { BODY("v_fract_f32 %0, %1") }
    if (WHICH == 27) { BODY("v_trunc_f32 %0, %1") }
    if (WHICH == 28) { BODY("v_cvt_u32_f32 %0, %1") }
    if (WHICH == 29) { BODY("v_alignbit_b32 %0, %0, %1, 31") }
    if (WHICH == 30) { BODY("v_perm_b32 %0, %0, %1, %2") }
    if (WHICH == 31) { BODY("v_mov_b32 %0, %1") }
    if (WHICH == 34) { BODY("v_and_or_b32 %0, %0, %1, %2") }
    if (WHICH == 35) { BODY("v_cmp_le_f32 vcc, 0, %0") }
    if (WHICH == 36) { BODY("v_mad_u32_u24 %0, %0, %1, %2") }
    if (WHICH == 37) { BODY("v_bfe_i32 %0, %1, 0, 16") }
    if (WHICH == 38) { BODY("v_lshl_or_b32 %0, %0, 3, %1") }
    if (WHICH == 39) { BODY("v_sub_f32_e64 %0, %8, |%1|") }
    if (WHICH == 40) { BODY("v_max_f32_e64 %0, |%0|, %1") }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + b0.x + b1.y + b2.x + b3.y;
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = t1 - t0;
        cyc[gridDim.x + blockIdx.x] = r1 - r0;   // s_memrealtime: constant 100 MHz
    }
}

template <int WHICH>
void run(const char* name, int wavesPerSimd)
{
    const int iters = 2000, ncu = 256;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)ncu * wavesPerSimd * 256 * 4);
    hipMalloc(&cyc, (size_t)ncu * wavesPerSimd * 8 * 2);
    hipLaunchKernelGGL(k<WHICH>, dim3(ncu * wavesPerSimd), dim3(256), 0, 0, 10, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WHICH>, dim3(ncu * wavesPerSimd), dim3(256), 0, 0, iters, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(ncu * wavesPerSimd * 2);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0, real = 0;
    const size_t nb = h.size() / 2;
    for (size_t i = 0; i < nb; ++i) avg += (double)h[i], real += (double)h[nb + i];
    avg /= nb;
    real /= nb;
    const double n = (double)iters * 32;  // instructions per wave
    // s_memtime ticks = shader cycles; all W waves of a SIMD run concurrently, so per-SIMD
    // throughput cost = wave cycles / (instructions of one wave * W)
    // shader clock while this loop ran = s_memtime ticks / s_memrealtime ticks x 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
    printf("%-28s W=%d  wave-cycles/instr %.2f  SIMD-cycles/instr %.2f  (kernel %.3f ms)  clock %.0f MHz  ns/instr/SIMD %.3f\n", name, wavesPerSimd, avg / n,
           avg / n / wavesPerSimd, ms, avg / real * 100.0, real * 10.0 / n / wavesPerSimd);
    hipFree(out);
    hipFree(cyc);
}

__global__ void k_round(const float* in, unsigned* out, int n)
{
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0);
}

int main()
{
    {
        float h[16] = {0.25f, 0.5f, 0.75f, 0.999f, 1.5f, 2.5f, 3.5f, 254.5f, 254.999f, 255.5f, 300.f, -0.25f, -0.75f, -3.f, 127.5f, 128.5f};
        float* d; unsigned* o; unsigned ho[16];
        hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
        hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_round, dim3(1), dim3(64), 0, 0, d, o, 16);
        hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) printf("cvt_pk_u8_f32(%g) = %u\n", h[i], ho[i]);
    }

    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w);
        run<13>("v_fmac_f32", w);
        run<11>("v_add_f32", w);
        run<12>("v_mul_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_pk_mul_f32", w);
        run<3>("v_pk_add_f32", w);
        run<4>("v_mul_f32_dpp quad", w);
        run<5>("v_fmac_f32_dpp quad", w);
        run<15>("v_fmac_f32_dpp half_mirror", w);
        run<6>("v_rndne_f32", w);
        run<7>("v_floor_f32", w);
        run<8>("v_cvt_pk_u8_f32", w);
        run<9>("v_cvt_f32_i32", w);
        run<16>("v_cvt_f32_i32_sdwa", w);
        run<10>("v_max_f32", w);
        run<14>("v_max3_f32", w);
        run<17>("v_cndmask_b32", w);
        run<18>("v_cmp_eq_f32", w);
        if (w == 4) {
            run<19>("v_and_b32", w); run<20>("v_lshlrev_b32", w); run<21>("v_add_u32", w); run<22>("v_sub_f32", w);
            run<23>("v_add_f32 |abs| e64", w); run<24>("v_fma_f32 |abs|", w); run<25>("v_min_f32", w); run<26>("v_fract_f32", w);
            run<27>("v_trunc_f32", w); run<28>("v_cvt_u32_f32", w); run<29>("v_alignbit_b32", w); run<30>("v_perm_b32", w);
            run<31>("v_mov_b32", w); run<34>("v_and_or_b32", w);
            run<35>("v_cmp_le_f32 0", w); run<36>("v_mad_u32_u24", w); run<37>("v_bfe_i32", w); run<38>("v_lshl_or_b32", w);
            run<39>("v_sub_f32 |abs| e64", w); run<40>("v_max_f32 |abs|", w);
            run<42>("v_cndmask_b32_e64 sgpr", w); run<43>("4x v_cndmask vcc rotating regs (per 4)", w); run<44>("v_bfi_b32", w);
            run<45>("v_cndmask_b32_e64 0,1", w); run<46>("v_ashrrev_i32", w); run<47>("v_min_u32", w); run<48>("cndmask+add pair (per 2)", w);
        }
        printf("\n");
    }
    return 0;
}
