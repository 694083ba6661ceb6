// libkpeg_amd/csrc/idct_colour.hip.h -- K4: dequantise + 8x8 IDCT + level shift + YCbCr->RGB
// + MCU tiling, for gfx950 (MI355X).
//
// Replaces MCU::constructMCU's dequantisation (src/MCU.cpp:110-120), MCU::computeIDCT
// (:172-216), performLevelShift (:218-245), convertYCbCrToRGB (:247-279) and
// Image::createImageFromMCUs (src/Image.cpp:20-86) of the reference.
//
// Bit-exactness.  The reference evaluates every sample as a 64-term sum accumulated in
// *float* in (u outer, v inner) order from double products; its rounding
// cannot be reproduced by a fast transform.  So:
//   (1) k_idct_colour_fast computes a fast separable f32 IDCT whose distance to the reference's float
//       result is bounded rigorously per block (tools/idct_bound.py derives the constant), and
//   (2) where the fast value lies within that bound of a rounding boundary (or the G term is too close
//       to an integer for the f32 colour arithmetic) it only MARKS the pixel in its tile loop and queues the
//       pixel's row in LDS; fx_flush, called when a wavefront's queue is full and when its tiles are done,
//       evaluates the marked pixels in the reference's own order and stores their bytes over the fast ones.
// Blocks with no AC coefficient are exact in (1) by construction.
//
// Mapping (no MFMA: byte/short work, not a dense contraction; the kernel is bound by its instruction stream).
//   * one workgroup of K4_WAVES wavefronts per CU; every wavefront an independent worker that takes tiles of
//     8 MCUs (64 x 8 pixels) from the workgroup's share through a counter in LDS; 8 lanes per MCU.
//   * lane j of an 8-lane group loads one 16-byte row of each component block
//     (rows 0,2,4,6 on lanes 0-3, rows 1,3,5,7 on lanes 4-7): a wavefront's three
//     global_load_dwordx4 cover 8 MCUs x 384 B = 3 KiB of contiguous coefficients -- or, compact stream, the
//     tile's records (4 bytes per non-zero AC coefficient) are scattered into an LDS image of the 24 blocks first.
//   * row pass (over v) in registers: even/odd decomposition, 34 f32 ops per 8 samples;
//     column pass (over u) across the 8 lanes with DPP: quad broadcasts feed 4-term
//     even (lanes 0-3) / odd (lanes 4-7) sums, one row_half_mirror FMA combines them.
//     Lane l ends up with pixel row l of the block: 8 pixels x 3 components.
//   * colour conversion in f32/int-exact arithmetic (proven ranges), RGB bytes staged in an
//     LDS tile (8 rows x 768 B, padded rows) and written back with 16-byte coalesced nontemporal stores.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kpeg_tables.h"

namespace kpeg_dev {

typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __constant__ double c_cos[64] = {
    0x1.0000000000000p+0, 0x1.f6297cff75cb0p-1,  0x1.d906bcf328d46p-1,  0x1.a9b66290ea1a3p-1,
    0x1.6a09e667f3bcdp-1, 0x1.1c73b39ae68c9p-1,  0x1.87de2a6aea964p-2,  0x1.8f8b83c69a60dp-3,
    0x1.0000000000000p+0, 0x1.a9b66290ea1a3p-1,  0x1.87de2a6aea964p-2,  -0x1.8f8b83c69a608p-3,
    -0x1.6a09e667f3bccp-1, -0x1.f6297cff75cb0p-1, -0x1.d906bcf328d47p-1, -0x1.1c73b39ae68c8p-1,
    0x1.0000000000000p+0, 0x1.1c73b39ae68c9p-1,  -0x1.87de2a6aea962p-2, -0x1.f6297cff75cb0p-1,
    -0x1.6a09e667f3bcep-1, 0x1.8f8b83c69a60cp-3,  0x1.d906bcf328d44p-1,  0x1.a9b66290ea1a5p-1,
    0x1.0000000000000p+0, 0x1.8f8b83c69a60dp-3,  -0x1.d906bcf328d46p-1, -0x1.1c73b39ae68c8p-1,
    0x1.6a09e667f3bcbp-1, 0x1.a9b66290ea1a5p-1,  -0x1.87de2a6aea965p-2, -0x1.f6297cff75cb2p-1,
    0x1.0000000000000p+0, -0x1.8f8b83c69a608p-3, -0x1.d906bcf328d47p-1, 0x1.1c73b39ae68c5p-1,
    0x1.6a09e667f3bcep-1, -0x1.a9b66290ea1a2p-1, -0x1.87de2a6aea971p-2, 0x1.f6297cff75cb0p-1,
    0x1.0000000000000p+0, -0x1.1c73b39ae68c6p-1, -0x1.87de2a6aea96dp-2, 0x1.f6297cff75cb0p-1,
    -0x1.6a09e667f3bc5p-1, -0x1.8f8b83c69a602p-3, 0x1.d906bcf328d46p-1,  -0x1.a9b66290ea1a1p-1,
    0x1.0000000000000p+0, -0x1.a9b66290ea1a4p-1, 0x1.87de2a6aea967p-2,  0x1.8f8b83c69a61dp-3,
    -0x1.6a09e667f3bc9p-1, 0x1.f6297cff75cb2p-1,  -0x1.d906bcf328d43p-1, 0x1.1c73b39ae68c2p-1,
    0x1.0000000000000p+0, -0x1.f6297cff75cb0p-1, 0x1.d906bcf328d44p-1,  -0x1.a9b66290ea1a2p-1,
    0x1.6a09e667f3bc4p-1, -0x1.1c73b39ae68c2p-1, 0x1.87de2a6aea95fp-2,  -0x1.8f8b83c69a616p-3};

// Quantiser tables in NATURAL (row-major) order, passed by value as a kernel argument.
struct QTables {
    uint16_t q[2][64];
};

struct IdctParams {
    const int16_t* coef;  // [mcu][3][8][8] quantised, natural order
    const float* ebound;  // [mcu][3] per-block error bound E (sample units), see k_ebound
    uint8_t* rgb;         // output stripe base (row 0 = first pixel row of mcu_row0)
    uint32_t mcus_w;      // MCUs per MCU row (width / 8)
    uint32_t mcu_rows;    // MCU rows to produce
    uint32_t pitch;       // bytes per pixel row (width * 3)
    uint32_t tiles_w;     // ceil(mcus_w / TILE_MCUS)
    uint32_t tiles_w_magic, tiles_w_shift;  // x / tiles_w == umulhi(x, magic) >> shift for x < 2^31 (host: div_magic)
    uint32_t ntiles;      // tiles_w * mcu_rows
    uint32_t* stats;      // [256] counters: [blockIdx & 255] += pixels sent to the exact path (may be null)
    uint32_t skip_exact;  // timing experiments only: count unsafe pixels but do not re-evaluate them
    uint32_t* status;     // the call's status words (device): stats == status + 16, [3] = wavefronts of this launch that are done
    uint32_t* h_status;   // host-pinned mirror (device address) or null: the launch's last wavefront copies the device words to it and clears them
    uint8_t* const* rgb_table;  // fused batch: output base of every image (device array), else null and rgb is the base
    uint32_t rows_per_img;      // ... MCU rows per image
    uint32_t keep_status; // batch lanes: leave the device words standing (error flags and counters accumulate over the lane's images)
    // k_idct_colour_fast<true>: the compact coefficient stream K2 writes (entropy.hip.h, WriteArgs) instead of `coef`.
    // A tile is 8 consecutive MCUs of the stream (mcus_w is a multiple of 8 on this path), its records are contiguous.
    const uint32_t* rec;          // [31:16] value, [13:8] natural position, [4:0] block within the tile (0..23)
    const int16_t* dc16;          // [blocks] absolute DC of every block
    const uint32_t* tile_start;   // [ntiles + 1] first record of every tile
    uint32_t rec_cap;             // records the buffer holds (bounds what a corrupt table can make a wavefront read)
};

constexpr uint32_t KPEG_STATUS_WORDS = 16 + 256 + 64 + 16;   // [1] error flags, [2] K1 passes, [3] + [272..335] end-of-call tickets, [16..271] counters, [336..351] K2 loop counts of KPEG_SYNC_STATS builds

// End of a call's last kernel, every wavefront: the last one to get here hands the status words to the host
// mirror (plain posted stores: no read over PCIe) and leaves the device COUNTERS zero for the next call -- no
// memset and no copy operation around a decode.  The error word [1] is sticky: it is never cleared here, so when
// several calls are enqueued before one kpeg_hip_sync() an earlier call's error flags are still standing on the
// device when a later call's epilogue copies them (kernels only OR into the word); kpeg_hip_sync() has the next
// call clear it.  keep: every word stays (a batch lane accumulates error flags and counters over its images;
// the mirror then always holds the sums so far).
// No fences (a release fence writes back the XCD's whole L2: 1280 of them tripled K4's time): every status update
// is a device-scope atomic performed at L2; `dep` is the value returned by this wavefront's own last update, so
// that update has been performed before the ticket is taken, and the last wavefront reads the words at L2.
__device__ __forceinline__ void status_epilogue(uint32_t* status, uint32_t* h_status, uint32_t nwaves, uint32_t keep, uint32_t dep)
{
    if (!h_status) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t last = 0;
    if (lane == 0) {
        // two levels (1280 tickets on one word would queue up for ~30 us at the kernel's tail): 64 slot words,
        // the wavefront that completes its slot takes one of 64 tickets on the top word
        const uint32_t slot = blockIdx.x & 63, in_slot = (nwaves - slot + 63) >> 6;
        if (atomicAdd(&status[272 + slot], 1u + (dep & 0u)) == in_slot - 1)
            last = atomicAdd(&status[3], 1u) == min(nwaves, 64u) - 1 ? 1u : 0u;
    }
    if (!__shfl((int)last, 0)) return;
    for (uint32_t w = lane; w < KPEG_STATUS_WORDS; w += 64) {
        const bool ticket = w == 3 || (w >= 272 && w < 336);
        const uint32_t v = ticket ? 0u : __hip_atomic_load(&status[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h_status[w] = v;
        if (ticket || (!keep && w != 1)) status[w] = 0;
    }
}

// ---- reference-order arithmetic ---------------------------------
// Compiled with -ffp-contract=off: each operation below is one IEEE operation.

// Cf[0] = (float)(1.0/sqrt(2.0)), Cf[k>0] = 1.0f; cc = Cf[u]*Cf[v] in float (MCU.cpp:189-192)
__device__ __forceinline__ float cc_of(int u, int v)
{
    const float c0 = 0x1.6a09e6p-1f;
    float cu = u == 0 ? c0 : 1.0f, cv = v == 0 ? c0 : 1.0f;
    return cu * cv;
}

// (int)roundl(ic) + 128 (MCU.cpp:228): round half away from zero, exact on a float
__device__ __forceinline__ int level_shift(float ic)
{
    float t = truncf(ic);
    float fr = ic - t;  // exact
    if (fr >= 0.5f) t += 1.0f;
    if (fr <= -0.5f) t -= 1.0f;
    // |ic| can exceed int range only for inputs far outside any JPEG; saturate like the
    // hardware conversion does instead of invoking UB.
    return (int)t + 128;
}

// convertYCbCrToRGB (MCU.cpp:255-265) on integer sample values, in double as written there
__device__ __forceinline__ uint32_t colour_exact(int sy, int scb, int scr)
{
    double Y = (double)(float)sy, Cb = (double)(float)scb, Cr = (double)(float)scr;
    int R = (int)floor(Y + 1.402 * (1.0 * Cr - 128.0));
    int G = (int)floor(Y - 0.344136 * (1.0 * Cb - 128.0) - 0.714136 * (1.0 * Cr - 128.0));
    int B = (int)floor(Y + 1.772 * (1.0 * Cb - 128.0));
    R = max(0, min(R, 255));
    G = max(0, min(G, 255));
    B = max(0, min(B, 255));
    return (uint32_t)R | ((uint32_t)G << 8) | ((uint32_t)B << 16);
}

// ---- mode 1: reference-order evaluation of every sample (cross-check kernel) -----------
// One 64-thread block per MCU, thread = pixel.  Slow by design.
__global__ __launch_bounds__(64) void k_idct_colour_exact(IdctParams p, QTables qt)
{
    __shared__ float s_fc[3][64];
    const uint32_t mcu = blockIdx.x;
    const int tid = threadIdx.x;
    const int u = tid >> 3, v = tid & 7;
    for (int c = 0; c < 3; ++c) {
        int q = qt.q[c ? 1 : 0][tid];
        int F = (int)p.coef[((size_t)mcu * 3 + c) * 64 + tid] * q;
        s_fc[c][tid] = cc_of(u, v) * (float)F;
    }
    __syncthreads();
    const int x = tid >> 3, y = tid & 7;
    int S[3];
    for (int c = 0; c < 3; ++c) {
        float sum = 0.0f;
        for (int k = 0; k < 64; ++k) {
            float fc = s_fc[c][k];
            if (fc != 0.0f) {
                double t = ((double)fc * c_cos[x * 8 + (k >> 3)]) * c_cos[y * 8 + (k & 7)];
                sum = (float)((double)sum + t);
            }
        }
        S[c] = level_shift((float)(0.25 * (double)sum));
    }
    uint32_t px = colour_exact(S[0], S[1], S[2]);
    uint32_t tr = mcu / p.mcus_w, tc = mcu % p.mcus_w;
    uint8_t* o = p.rgb + (size_t)(tr * 8 + x) * p.pitch + (size_t)(tc * 8 + y) * 3;
    o[0] = (uint8_t)px;
    o[1] = (uint8_t)(px >> 8);
    o[2] = (uint8_t)(px >> 16);
}

// ---- 4:2:0 (extension): reference-order evaluation of every sample of a 16x16 MCU --------
// One 256-thread block per MCU.  coef: [mcu][Y00 Y01 Y10 Y11 Cb Cr][64] natural order; every thread evaluates one luma
// sample, the first 128 also one chroma sample each, in the reference's own order (as k_idct_colour_exact); then thread
// (py, px) converts its pixel with the chroma sample that covers it (each repeated 2x2: no interpolation, no arithmetic
// the reference does not have).  rgb is the picture padded to whole MCUs (pitch bytes per row); the caller crops.
__global__ __launch_bounds__(256) void k_idct_colour_exact_420(const int16_t* __restrict__ coef, uint8_t* __restrict__ rgb, uint32_t mcus_w,
                                                              uint32_t pitch, QTables qt)
{
    __shared__ float s_fc[6][64];
    __shared__ int s_S[6][64];
    const uint32_t mcu = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < 384; i += 256) {
        const int blk = i >> 6, k = i & 63;
        const int F = (int)coef[(size_t)mcu * 384 + i] * (int)qt.q[blk < 4 ? 0 : 1][k];
        s_fc[blk][k] = cc_of(k >> 3, k & 7) * (float)F;
    }
    __syncthreads();
    for (int i = tid; i < 384; i += 256) {
        const int blk = i >> 6, x = (i >> 3) & 7, y = i & 7;
        float sum = 0.0f;
        for (int k = 0; k < 64; ++k) {
            const float fc = s_fc[blk][k];
            if (fc != 0.0f) {
                const double t = ((double)fc * c_cos[x * 8 + (k >> 3)]) * c_cos[y * 8 + (k & 7)];
                sum = (float)((double)sum + t);
            }
        }
        s_S[blk][x * 8 + y] = level_shift((float)(0.25 * (double)sum));
    }
    __syncthreads();
    const int py = tid >> 4, px = tid & 15;
    const uint32_t p = colour_exact(s_S[(py >> 3) * 2 + (px >> 3)][(py & 7) * 8 + (px & 7)], s_S[4][(py >> 1) * 8 + (px >> 1)],
                                    s_S[5][(py >> 1) * 8 + (px >> 1)]);
    const uint32_t tr = mcu / mcus_w, tc = mcu % mcus_w;
    uint8_t* o = rgb + (size_t)(tr * 16 + py) * pitch + (size_t)(tc * 16 + px) * 3;
    o[0] = (uint8_t)p;
    o[1] = (uint8_t)(p >> 8);
    o[2] = (uint8_t)(p >> 16);
}

// ---- mode 0: fast path + exact re-evaluation ---------------------------------------------

// Error bound (derivation and numeric check: tools/idct_bound.py, DESIGN.md "K4 exactness"):
//   |fast - reference float result| <= E = U * A * (nnz_ac + KAPPA)            (sample units)
// U = 2^-24 (1 + 2^-10), A = sum |in| over the block (in = 0.25 * cc * Q * coefficient),
// nnz_ac = non-zero AC coefficients (one float rounding of the reference's accumulator per
// non-zero AC term), KAPPA bounds the fast path's own roundings.  Blocks without AC terms are
// exact (E = 0).  E is produced per block by whoever writes the coefficients (K2, or k_ebound
// for caller-supplied coefficients) and read here as a 4-byte sidecar per block.
#define KPEG_KAPPA 14.5f      // tools/idct_bound.py prints 14.444
#define KPEG_U 0x1.004p-24f   // 2^-24 (1 + 2^-10): covers the f32 rounding of A's own summation
// |t - rint(t)| below this sends the G channel to the exact path (f32 error of t <= 3.7e-5)
#define KPEG_G_DELTA 6.0e-5f

// Range guards folded into the bound: every fast sample satisfies |v| <= A (1 + 2^-20).
//   A >= KPEG_A_LIM (any component): E = +inf, all the block's samples take the reference-order path.  Below it every sample
//       is an integer of at most 2041 in magnitude once rounded: inside the range the colour arithmetic's rounding argument is
//       checked for (|sample| <= 4100, tests/test_tables.py) and exact as an f16 (K4's row queue keeps the rounded samples so).
//   A >= KPEG_A_LIM_CHROMA (chroma block): the samples may leave the range the f32 colour arithmetic is proven
//       for (|.| < 250): the lowest mantissa bit of E is set (E is first rounded up to an even mantissa, so the
//       bit never shrinks it) and K4 converts that MCU's pixels with the reference's double arithmetic in-lane.
//       Saturated colour edges do this in photographs; dense noise does it everywhere.
#define KPEG_A_LIM_CHROMA 249.0f
#define KPEG_A_LIM 2040.0f
// The sign bit carries one more fact about the block: set = every non-zero AC coefficient sits at
// (0,1), (1,0) or (1,1).  Those blocks produce nearly all true ties (equal and opposite (0,1)/(1,0)
// terms cancel on the diagonal).  (K4 reads the magnitude only; the sign is kept for tools and tests.)
__device__ __forceinline__ float block_ebound(float A, int nnz_ac, bool chroma, bool corner_only)
{
    if (!(A < KPEG_A_LIM)) return __builtin_inff();
    const float E = nnz_ac ? (KPEG_U * A) * ((float)nnz_ac + KPEG_KAPPA) : 0.0f;
    uint32_t bits = (__float_as_uint(E) + 1u) & ~1u;
    if (chroma && !(A < KPEG_A_LIM_CHROMA)) bits |= 1u;
    const float Ef = __uint_as_float(bits);
    return corner_only ? -Ef : Ef;
}

// E for caller-supplied coefficients (kpeg_hip_idct_colour): eight lanes per block, one 16-byte coefficient row each -- a wavefront
// reads 1 KiB of consecutive coefficients per load (a thread per block, rounds 1-3, read 16 bytes of 64 different lines per load:
// 0.20 ms for the 8K image's 1.5 M blocks, three times K4 itself) -- and the rows' sums meet over the eight lanes.  (A's terms are
// added in another order than K2 adds them: KPEG_U's slack covers any order.)
constexpr uint32_t EB_BLOCKS = 32;   // blocks per workgroup of 256 threads
__global__ __launch_bounds__(256) void k_ebound(const int16_t* coef, uint32_t nblocks, QTables qt, float* ebound)
{
    __shared__ float s_w[2][64];   // 0.25 * cc * Q per position (the DC term's own chain apart)
    if (threadIdx.x < 128) {
        const int t = threadIdx.x >> 6, k = threadIdx.x & 63;
        s_w[t][k] = 0.25f * cc_of(k >> 3, k & 7) * (float)qt.q[t][k];
    }
    __syncthreads();
    const uint32_t b = blockIdx.x * EB_BLOCKS + (threadIdx.x >> 3);
    const int r = threadIdx.x & 7;
    const bool live = b < nblocks;
    const int t = (b % 3) ? 1 : 0;
    uint4 d = make_uint4(0, 0, 0, 0);
    if (live) d = reinterpret_cast<const uint4*>(coef + (size_t)b * 64)[r];
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
    float A = 0.f;
    int n = 0;
    bool corner = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (int)(short)((w[i >> 1] >> ((i & 1) * 16)) & 0xFFFF);
        const int k = r * 8 + i;
        if (k == 0) {
            A += fabsf(0.25f * (cc_of(0, 0) * ((float)c * (float)qt.q[t][0])));
        } else if (c != 0) {
            A += fabsf((float)c * s_w[t][k]);
            n++;
            corner = corner && (k == 1 || k == 8 || k == 9);
        }
    }
    int cbit = corner ? 1 : 0;
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        A += __shfl_xor(A, m);
        n += __shfl_xor(n, m);
        cbit &= __shfl_xor(cbit, m);
    }
    if (live && r == 0) ebound[b] = block_ebound(A, n, t != 0, cbit != 0);
}

typedef unsigned int uint3v __attribute__((ext_vector_type(3)));
typedef unsigned int __attribute__((ext_vector_type(4), may_alias)) uint4v;  // 16-byte view of uint32_t LDS words
constexpr int TILE_MCUS = 8;                    // MCUs per wavefront iteration (8 lane groups)
constexpr int TILE_ROW_BYTES = TILE_MCUS * 24;  // 192
constexpr int TILE_ROW_STRIDE = 208;            // padded: 13 x 16 bytes, conflict-free b64 writes across rows
#ifndef KPEG_K4_WAVES
#define KPEG_K4_WAVES 16
#endif
// K4's workgroup: as many wavefronts as a CU is to hold (4 SIMDs x 4), ONE workgroup per CU.  The wavefronts are independent
// workers (no barrier after the start-up); what they share is the LDS counter that hands out the workgroup's tiles.
constexpr int K4_WAVES = KPEG_K4_WAVES, K4_THREADS = 64 * K4_WAVES;

// 1-D 8-point inverse DCT kernel sum_v a[v] cos((2y+1) v pi/16), y = 0..7, in place.
// NV < 8: a[NV..7] are known to be zero and their terms are left out.  fma(0, c, x) == x and
// x + 0 == x exactly, so the pruned forms return the same floats as the full one (tools/idct_bound.py
// analyses the full sequence; its bound covers them).
template <int NV>
__device__ __forceinline__ void row_idct8(float a[8])
{
    const float c1 = 0.98078528040323044913f, c2 = 0.92387953251128675613f, c3 = 0.83146961230254523708f,
                c4 = 0.70710678118654752440f, c5 = 0.55557023301960222474f, c6 = 0.38268343236508977173f,
                c7 = 0.19509032201612826785f;
    float t0 = NV > 4 ? __builtin_fmaf(a[4], c4, a[0]) : a[0];
    float t1 = NV > 4 ? __builtin_fmaf(a[4], -c4, a[0]) : a[0];
    float e0, e1, e2, e3, o0, o1, o2, o3;
    if (NV > 2) {
        float p = NV > 6 ? __builtin_fmaf(a[6], c6, a[2] * c2) : a[2] * c2;
        float q = NV > 6 ? __builtin_fmaf(a[6], -c2, a[2] * c6) : a[2] * c6;
        e0 = t0 + p, e3 = t0 - p, e1 = t1 + q, e2 = t1 - q;
        o0 = __builtin_fmaf(a[3], c3, a[1] * c1);
        o1 = __builtin_fmaf(a[3], -c7, a[1] * c3);
        o2 = __builtin_fmaf(a[3], -c1, a[1] * c5);
        o3 = __builtin_fmaf(a[3], -c5, a[1] * c7);
    } else {
        // a[2..7] == 0: p = q = +0 and the fma's addend terms are +-0, which change no value (x + 0 == x)
        e0 = e1 = e2 = e3 = t0;
        o0 = a[1] * c1, o1 = a[1] * c3, o2 = a[1] * c5, o3 = a[1] * c7;
    }
    if (NV > 5) {
        o0 = __builtin_fmaf(a[5], c5, o0);
        o1 = __builtin_fmaf(a[5], -c1, o1);
        o2 = __builtin_fmaf(a[5], c7, o2);
        o3 = __builtin_fmaf(a[5], c3, o3);
    }
    if (NV > 7) {
        o0 = __builtin_fmaf(a[7], c7, o0);
        o1 = __builtin_fmaf(a[7], -c5, o1);
        o2 = __builtin_fmaf(a[7], c3, o2);
        o3 = __builtin_fmaf(a[7], -c1, o3);
    }
    a[0] = e0 + o0;
    a[7] = e0 - o0;
    a[1] = e1 + o1;
    a[6] = e1 - o1;
    a[2] = e2 + o2;
    a[5] = e2 - o2;
    a[3] = e3 + o3;
    a[4] = e3 - o3;
}

// Column pass across the 8 lanes of an MCU group, all 8 pixel columns at once.
// Lanes 0-3 hold rows 0,2,4,6 of g and produce the even sums, lanes 4-7 hold rows 1,3,5,7 and
// produce the negated odd sums; quad_perm broadcasts feed v_fmac_f32_dpp directly (hipcc only
// folds DPP into VOP2 multiplies, not into FMAs, hence the asm), and one row_half_mirror FMA
// combines the two halves: out = own + mirror * s.
// Hazard (VALU write -> DPP read of the same VGPR needs 2 wait states): the leading s_nop covers
// the compiler-produced g; inside, every DPP read is >= 8 instructions behind its producer.
// NU < 8: coefficient rows NU..7 are zero in every block of the wavefront, i.e. g == 0 on the lanes
// that hold them; their broadcast terms (quad lanes NU/2..3) are left out: fmac(acc, 0, k) == acc.
template <int NU>
__device__ __forceinline__ void column_idct8(const float g[8], float k0, float k1, float k2, float k3, float s, float o[8])
{
#define KPEG_DPP8(op, sel, kreg)                                                                    \
    op " %0, %8, " kreg " " sel " row_mask:0xf bank_mask:0xf\n\t"                                   \
    op " %1, %9, " kreg " " sel " row_mask:0xf bank_mask:0xf\n\t"                                   \
    op " %2, %10, " kreg " " sel " row_mask:0xf bank_mask:0xf\n\t"                                  \
    op " %3, %11, " kreg " " sel " row_mask:0xf bank_mask:0xf\n\t"                                  \
    op " %4, %12, " kreg " " sel " row_mask:0xf bank_mask:0xf\n\t"                                  \
    op " %5, %13, " kreg " " sel " row_mask:0xf bank_mask:0xf\n\t"                                  \
    op " %6, %14, " kreg " " sel " row_mask:0xf bank_mask:0xf\n\t"                                  \
    op " %7, %15, " kreg " " sel " row_mask:0xf bank_mask:0xf\n\t"
    if (NU > 6) {
    asm("s_nop 1\n\t"
        KPEG_DPP8("v_mul_f32_dpp", "quad_perm:[0,0,0,0]", "%16")
        KPEG_DPP8("v_fmac_f32_dpp", "quad_perm:[1,1,1,1]", "%17")
        KPEG_DPP8("v_fmac_f32_dpp", "quad_perm:[2,2,2,2]", "%18")
        KPEG_DPP8("v_fmac_f32_dpp", "quad_perm:[3,3,3,3]", "%19")
        "v_fmac_f32_dpp %0, %0, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %4, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %5, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %6, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %7, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
        : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(k0), "v"(k1),
          "v"(k2), "v"(k3), "v"(s));
    } else if (NU > 4) {
    asm("s_nop 1\n\t"
        KPEG_DPP8("v_mul_f32_dpp", "quad_perm:[0,0,0,0]", "%16")
        KPEG_DPP8("v_fmac_f32_dpp", "quad_perm:[1,1,1,1]", "%17")
        KPEG_DPP8("v_fmac_f32_dpp", "quad_perm:[2,2,2,2]", "%18")
        "v_fmac_f32_dpp %0, %0, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %4, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %5, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %6, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %7, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
        : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(k0), "v"(k1),
          "v"(k2), "v"(k3), "v"(s));
    } else if (NU <= 2) {
    asm("s_nop 1\n\t"
        KPEG_DPP8("v_mul_f32_dpp", "quad_perm:[0,0,0,0]", "%16")
        "v_fmac_f32_dpp %0, %0, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %4, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %5, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %6, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %7, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
        : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(k0), "v"(k1),
          "v"(k2), "v"(k3), "v"(s));
    } else {
    asm("s_nop 1\n\t"
        KPEG_DPP8("v_mul_f32_dpp", "quad_perm:[0,0,0,0]", "%16")
        KPEG_DPP8("v_fmac_f32_dpp", "quad_perm:[1,1,1,1]", "%17")
        "v_fmac_f32_dpp %0, %0, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %4, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %5, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %6, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %7, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
        : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(k0), "v"(k1),
          "v"(k2), "v"(k3), "v"(s));
    }
#undef KPEG_DPP8
}

__host__ __device__ constexpr float cosf_tab(int k)  // cos(k*pi/16), k = 0..31, f32-rounded
{
    const float t[9] = {1.0f,
                        0.98078528040323044913f,
                        0.92387953251128675613f,
                        0.83146961230254523708f,
                        0.70710678118654752440f,
                        0.55557023301960222474f,
                        0.38268343236508977173f,
                        0.19509032201612826785f,
                        0.0f};
    k &= 31;
    if (k > 16) k = 32 - k;
    return k <= 8 ? t[k] : -t[16 - k];
}

struct LaneConst {
    float q0[2];     // Q[u][0] as float (exact DC-column chain)
    float cc0;       // cc[u][0]
    float k[4];      // column-pass constants
    float s;         // -1 on even-row lanes, +1 on odd-row lanes
};

// One component block: d = 8 int16 (row u of the block, this lane's share).
// out[i] = fast value of sample i of pixel row (lane & 7).
// N < 8: the block's coefficients outside its top-left N x N corner are zero, for every block this
// wavefront holds of this component (wave-uniform choice: see the caller).
template <int N>
__device__ __forceinline__ void block_fast(const uint4 d, const LaneConst& lc, const float* __restrict__ m, int tab, float out[8])
{
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
    float a[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[2 * i] = 2 * i < N ? (float)(short)(w[i] & 0xFFFF) : 0.0f;
        a[2 * i + 1] = 2 * i + 1 < N ? (float)((int)w[i] >> 16) : 0.0f;
    }
    // column 0 through the reference's own chain 0.25 * (cc * (float)(c*Q)): exact for the DC term
    a[0] = 0.25f * (lc.cc0 * (a[0] * lc.q0[tab]));
    // AC input scale 0.25 * cc[u][v] * Q[u][v] of this lane's row, from LDS (m[0] unused)
    const float4 mlo = *reinterpret_cast<const float4*>(m);
    a[1] *= mlo.y;
    if (N > 2) {
        a[2] *= mlo.z;
        a[3] *= mlo.w;
    }
    if (N > 4) {
        const float4 mhi = *reinterpret_cast<const float4*>(m + 4);
        a[4] *= mhi.x;
        a[5] *= mhi.y;
        if (N > 6) {
            a[6] *= mhi.z;
            a[7] *= mhi.w;
        }
    }
    row_idct8<N>(a);
    column_idct8<N>(a, lc.k[0], lc.k[1], lc.k[2], lc.k[3], lc.s, out);
}

__device__ __forceinline__ uint32_t tile_row(const IdctParams& p, uint32_t tile)
{
    return p.tiles_w == 1 ? tile : (__umulhi(tile, p.tiles_w_magic) >> p.tiles_w_shift);
}

__device__ __forceinline__ uint32_t pk_u8(float v, uint32_t sel, uint32_t old)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(v, sel, old);  // saturating float -> byte `sel` of old
}

// ---- reference-order evaluation of single samples -------------------------------------------------
// One sample of any block, evaluated by the whole wavefront: lane p owns coefficient position p = u*8+v
// (row-major = the reference's loop order) and computes its product term; the float accumulation then walks
// the non-zero lanes in order.  fc: cc * (float)(coefficient * Q) of this lane's position; x, y wave-uniform.
// Returns (int)roundl(ic) + 128 in every lane.
__device__ __forceinline__ int exact_sample_wave(float fc, const double* __restrict__ s_cos, int x, int y, bool nz)
{
    const int lane = __lane_id();
    const int u = lane >> 3, v = lane & 7;
    const double t = ((double)fc * s_cos[x * 8 + u]) * s_cos[y * 8 + v];
    unsigned long long live = __ballot(nz);
    float sum = 0.0f;
    while (live) {
        const int p = __builtin_ctzll(live);
        live &= live - 1;
        // p is wave-uniform: two v_readlane_b32
        const long long tb = __builtin_bit_cast(long long, t);
        const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)tb, p);
        const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(tb >> 32), p);
        const double tp = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        sum = (float)((double)sum + tp);
    }
    return level_shift((float)(0.25 * (double)sum));
}

// One sample of any block, evaluated by one lane: the 64-term sum in the reference's order (u outer, v inner), zero
// coefficients skipped (they leave the float accumulator unchanged).  Used when many samples are due at once (a tile
// evaluated as a whole): every lane of a pass holds a different sample; rows that are zero on every lane are skipped.
//   blk: the block's eight coefficient rows in global memory (natural order); qi: the component's quantisers (LDS).
__device__ __forceinline__ int exact_sample_lane(const uint4* __restrict__ blk, const uint32_t* __restrict__ qi,
                                                 const double* __restrict__ s_cos, int x, int y)
{
    float sum = 0.0f;
    // two halves of four rows, each half's loads in flight together (all eight would need more registers than the
    // tile loop leaves; one after the other is eight memory latencies)
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        uint4 rows[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) rows[k] = blk[h * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 d = rows[k];
            if (__ballot((d.x | d.y | d.z | d.w) != 0) == 0) continue;  // wave-uniform
            const int u = h * 4 + k;
            const uint32_t w[4] = {d.x, d.y, d.z, d.w};
            const double cxu = s_cos[x * 8 + u];
            const float cu = u == 0 ? 0x1.6a09e6p-1f : 1.0f;
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int cf = (v & 1) ? ((int)w[v >> 1] >> 16) : (int)(short)(w[v >> 1] & 0xFFFF);
                if (cf != 0) {
                    const int F = cf * (int)qi[u * 8 + v];                     // m_8x8block after MCU.cpp:110-112
                    const float cc = cu * (v == 0 ? 0x1.6a09e6p-1f : 1.0f);    // Cf[u] * Cf[v] in float (cc_of)
                    const float fc = cc * (float)F;                            // float multiply (MCU.cpp:189-192)
                    const double t = ((double)fc * cxu) * s_cos[y * 8 + v];    // two double multiplies
                    sum = (float)((double)sum + t);                            // float accumulator
                }
            }
        }
    }
    return level_shift((float)(0.25 * (double)sum));
}

constexpr int IMG_BYTES = 24 * 128;       // compact path: the tile's 24 blocks rebuilt in LDS, natural order, int16

// ---- the marked pixels, in the reference's own order ----------------------------------------------------------------------
// K4's tile loop only MARKS the pixels whose fast value cannot be trusted.  A tile that has any (two in five on the 8K workload)
// appends the pixel rows that hold them to the wavefront's queue in LDS -- a row entry carries all that settling it takes, so
// that nothing has to be fetched again -- and when the queue may not hold another tile's rows (or the wavefront's tiles are
// done) fx_flush settles what has gathered, ONE LANE PER ROW, a marked pixel of every row per round: the components the tile
// loop does not vouch for are evaluated as MCU::computeIDCT has them (MCU.cpp:184-198: float accumulator, double products,
// u outer, v inner, zero terms leave the accumulator as it is), the others keep the rounded fast value, then performLevelShift
// and convertYCbCrToRGB (colour_exact), and the pixel's three bytes are stored over what the tile loop wrote.
// Why a queue across tiles: a marked tile has six marked pixels on average.  Settling them tile by tile runs the whole
// reference-order code (some 300 instructions, doubles) for six lanes of 64 -- measured 24-27 us of K4's 83 (profiles/r03_d) --
// whereas a full queue keeps most lanes busy: a few passes in a wavefront's life.
//   corner-only blocks (the sign of the block's bound; nine in ten: near-ties of chroma blocks, equal and opposite (0,1)/(1,0)
//     terms that cancel on the diagonal): four terms by the lane itself, from two words of the block (part of the entry);
//   any other block, compact stream: settled when the row is queued, while the tile's rebuilt blocks stand in LDS
//     (fx_noncorner) -- the sample's value replaces the fast one in the entry and the entry vouches for it;
//   any other block, dense layout: by the lane itself in fx_flush, 64 terms in order from the block's rows in memory.
// Neither function is inlined: they run a few times in a wavefront's life, and their registers are not to be paid for by the
// tile loop, which has none to spare (56 bytes of spill per lane, an inlined version's, took K4 from 55 to 85 us).  A called
// function waits for everything the wavefront has in flight (s_waitcnt 0 at its entry): paid a few times, not per tile.
constexpr int ROWQ_CAP = 64;      // rows a wavefront's queue holds: a tile's worth
constexpr int ROWQ_WORDS = 20;    // per row: [0, 12) [component][pixel column] the rounded fast sample (minus the level shift) as f16 (exact: a block
                                  // with finite bound has |sample| <= KPEG_A_LIM < 2048); [12] per component Y Cb Cr and for the G term one byte,
                                  // bit 7 - i set = pixel column i is vouched for; [13] tile << 6 | lane of the tile loop (MCU of the tile << 3 |
                                  // pixel row); [14, 20) compact stream: words 0 and 4 of the MCU's three blocks = coefficients (0,0),(0,1) and (1,0),(1,1)
struct FxArgs {   // what fx_flush needs of IdctParams, by value (a reference to the kernel's arguments would put them on the stack)
    uint8_t* rgb;
    uint8_t* const* rgb_table;
    uint32_t pitch, tiles_w, tiles_w_magic, tiles_w_shift, ntiles, rows_per_img;
};
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef __attribute__((address_space(3))) const double lds_cf64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;

// Value of one sample of any block, evaluated by the whole wavefront (exact_sample_wave with its table in LDS address space)
__device__ __forceinline__ int fx_sample_wave(float fc, lds_cf64* s_cos, int x, int y, bool nz)
{
    const int lane = __lane_id();
    const int u = lane >> 3, v = lane & 7;
    const double t = ((double)fc * s_cos[x * 8 + u]) * s_cos[y * 8 + v];
    unsigned long long live = __ballot(nz);
    float sum = 0.0f;
    while (live) {
        const int p = __builtin_ctzll(live);
        live &= live - 1;
        const long long tb = __builtin_bit_cast(long long, t);
        const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)tb, p);
        const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(tb >> 32), p);
        const double tp = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        sum = (float)((double)sum + tp);
    }
    return level_shift((float)(0.25 * (double)sum));
}

// Compact stream, when a tile's rows have just been queued: the marked samples of blocks that are NOT corner-only, evaluated while
// the tile's blocks stand rebuilt in LDS (img: [24][64] int16, natural order).  nc: the lane's row's samples concerned, one byte per
// component, bit 7 - i = pixel column i (0: none); entry: the lane's queue entry.  The value goes into the entry in place of the
// fast one (exact in f16: a block with a finite bound has |sample| <= KPEG_A_LIM), and the entry then vouches for it -- and no
// longer for the pixel's G term, so that the pixel stays marked whatever else it has (fx_flush converts every marked pixel with
// the reference's own colour arithmetic anyway).
//   fx_noncorner_few: a few samples in the tile (nearly always): the whole wavefront takes them one after the other, lane =
//     coefficient position.  Inlined: a call would wait for everything the wavefront has in flight, the next tile's coefficients
//     and the previous tile's pixels -- 22 us of K4 when every tenth tile did (profiles/r03_d).
//   fx_noncorner_many (a call): many (a cluster of ties): every lane its own, 64 terms in order; and the rows of MCUs that have
//     a block whose bound is infinite (inf: bit c; its samples may be anything, f16 does not hold them): the lane finishes its
//     whole pixel row here, every component by the 64-term sum, into the wavefront's finished tile in LDS (`trow`: the lane's 24
//     bytes of it), which has not been written back yet; the entry then vouches for everything and fx_flush passes it over.
__device__ __forceinline__ void fx_vouch(lds_u32* entry, uint32_t vouch)
{
    // bits [23:0]: samples settled; the same pixels' G bits are cleared
    if (vouch) entry[12] = (entry[12] | vouch) & ~(((vouch | (vouch >> 8) | (vouch >> 16)) & 0xFFu) << 24);
}

__device__ __forceinline__ void fx_noncorner_few(lds_u32* entry, lds_cu32* img, lds_cu32* s_qi, lds_cf64* s_cos, uint32_t nc)
{
    const uint32_t lane = (uint32_t)__lane_id();
    lds_u16* const eh = reinterpret_cast<lds_u16*>(entry);
    const float ccl = cc_of((int)(lane >> 3), (int)(lane & 7));
    uint32_t vouch = 0;
    unsigned long long rows = __ballot(nc != 0);
    while (rows) {   // wave-uniform
        const int L = __builtin_ctzll(rows);
        rows &= rows - 1;
        uint32_t ncL = (uint32_t)__builtin_amdgcn_readlane((int)nc, L);
        while (ncL) {
            const uint32_t b = 31u - (uint32_t)__builtin_clz(ncL);
            ncL &= ~(1u << b);
            const uint32_t c = b >> 3, y = 7u - (b & 7u);
            const uint32_t wv = img[(((uint32_t)L >> 3) * 3u + c) * 32u + (lane >> 1)];
            const int cf = (lane & 1u) ? ((int)wv >> 16) : (int)(short)(wv & 0xFFFF);
            const int F = cf * (int)s_qi[(c ? 64 : 0) + lane];   // m_8x8block after MCU.cpp:110-112
            const int S = fx_sample_wave(ccl * (float)F, s_cos, L & 7, (int)y, F != 0);
            if ((int)lane == L) {
                eh[c * 8 + y] = __builtin_bit_cast(uint16_t, (_Float16)(float)(S - 128));
                vouch |= 1u << b;
            }
        }
    }
    fx_vouch(entry, vouch);
}

#ifdef KPEG_FX_MANY_INLINE
#define KPEG_FX_MANY_ATTR __forceinline__
#else
#define KPEG_FX_MANY_ATTR __attribute__((noinline))
#endif
__device__ KPEG_FX_MANY_ATTR void fx_noncorner_many(lds_u32* entry, lds_cu32* img, lds_cu32* s_qi, lds_cf64* s_cos, uint32_t nc, uint32_t inf,
                                                            __attribute__((address_space(3))) uint8_t* trow)
{
    const uint32_t lane = (uint32_t)__lane_id();
    const float c0 = 0x1.6a09e6p-1f;
    // MCU::computeIDCT's sum for sample (x, y) of block c of MCU g of the tile, in its order (MCU.cpp:184-198)
    auto sum64 = [&](uint32_t g, uint32_t x, uint32_t c, uint32_t y) -> int {
        lds_cf64* const cx = s_cos + x * 8;
        lds_cf64* const cy = s_cos + y * 8;
        const __attribute__((address_space(3))) uint4v* const rows = reinterpret_cast<const __attribute__((address_space(3))) uint4v*>(img + (g * 3u + c) * 32u);
        lds_cu32* const qi = s_qi + (c ? 64 : 0);
        float sum = 0.0f;
#pragma unroll 1
        for (int uu = 0; uu < 8; ++uu) {
            const uint4v d = rows[uu];
            if ((d.x | d.y | d.z | d.w) == 0u) continue;   // (quantised blocks are mostly empty rows; a zero term leaves the float accumulator unchanged)
            const uint32_t w[4] = {d.x, d.y, d.z, d.w};
            const double cxu = cx[uu];
            const float cu = uu == 0 ? c0 : 1.0f;
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int cf = (v & 1) ? ((int)w[v >> 1] >> 16) : (int)(short)(w[v >> 1] & 0xFFFF);
                if (cf != 0) {
                    const int F = cf * (int)qi[uu * 8 + v];                     // m_8x8block after MCU.cpp:110-112
                    const float fc = (cu * (v == 0 ? c0 : 1.0f)) * (float)F;    // Cf[u] * Cf[v] in float, float multiply (MCU.cpp:189-192)
                    const double t = ((double)fc * cxu) * cy[v];
                    sum = (float)((double)sum + t);
                }
            }
        }
        return level_shift((float)(0.25 * (double)sum));
    };
    if (__ballot(inf != 0)) {   // wave-uniform; hostile or broken streams
        if (inf) {
            const uint32_t g = lane >> 3, x = lane & 7u;   // MCU of the tile, pixel row
#pragma unroll 1
            for (uint32_t y = 0; y < 8; ++y) {
                int S[3];
#pragma unroll 1
                for (uint32_t c = 0; c < 3; ++c) S[c] = sum64(g, x, c, y);
                const uint32_t px = colour_exact(S[0], S[1], S[2]);
                trow[y * 3] = (uint8_t)px, trow[y * 3 + 1] = (uint8_t)(px >> 8), trow[y * 3 + 2] = (uint8_t)(px >> 16);
            }
            entry[12] = 0xFFFFFFFFu;
            nc = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // The samples are dealt to the lanes one by one, whatever rows they sit in: a cluster of ties puts up to 24 samples in ONE pixel
    // row, and with a lane per row (round 3's first build) that lane's sums ran one after the other while 63 lanes looked on -- up to
    // 14.5 us on one tile, and the wavefronts that met such tiles last were the kernel's tail (profiles/r03_k_*).  Sample number i of
    // the tile (rows in lane order, a row's samples from its highest bit down) goes to lane i mod 64: the row it belongs to is found by
    // a binary search over the rows' running counts (ds_bpermute: no memory), its value goes to that row's entry.
    const uint32_t cnt = (uint32_t)__popc(nc), incl = wave_scan_incl(cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t entry_addr = (uint32_t)(uintptr_t)entry;
    for (uint32_t base = 0; base < total; base += 64) {   // wave-uniform
        const uint32_t i = base + lane;
        const bool act = i < total;
        uint32_t owner = 0;   // rows whose running count is <= i
#pragma unroll
        for (uint32_t s = 32; s; s >>= 1) {
            const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((owner + s - 1) << 2), (int)incl);
            owner += v <= i ? s : 0u;
        }
        owner = act ? owner : 63u;
        uint32_t kth = i - (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)(incl - cnt));   // the row's kth sample, from the highest bit down
        uint32_t bits = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)nc);
        const uint32_t eaddr = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)entry_addr);
        kth = act ? kth : 0u;
        while (__ballot(kth != 0)) {   // wave-uniform: at most 23 rounds of three instructions
            if (kth) {
                bits &= ~(1u << (31u - (uint32_t)__builtin_clz(bits)));
                --kth;
            }
        }
        if (act && bits) {
            const uint32_t b = 31u - (uint32_t)__builtin_clz(bits);
            const uint32_t c = b >> 3, y = 7u - (b & 7u);
            const int S = sum64(owner >> 3, owner & 7u, c, y);
            lds_u32* const eo = (lds_u32*)(uintptr_t)eaddr;
            reinterpret_cast<lds_u16*>(eo)[c * 8 + y] = __builtin_bit_cast(uint16_t, (_Float16)(float)(S - 128));
            // the entry vouches for the sample and no longer for the pixel's G term (fx_vouch, one bit at a time: several lanes may
            // hold samples of one row)
            __hip_atomic_fetch_or(eo + 12, 1u << b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_and(eo + 12, ~(1u << (24u + (b & 7u))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// q: the wavefront's queue, nrows entries; s_qi: [2][64] quantisers, natural order.  Reads LDS only (an entry carries all it takes);
// what is still unsafe in an entry belongs to a corner-only block (fx_noncorner_* have settled the rest).
__device__ __attribute__((noinline)) void fx_flush(FxArgs a, lds_cu32* q, uint32_t nrows, lds_cu32* s_qi, lds_cf64* s_cos)
{
    // ONE LANE PER MARKED PIXEL (the round's first build gave a lane a row and went round as often as the fullest row had marked
    // pixels, every round through all three components' code): the queue's marked pixels are numbered in row order, a row's from
    // its highest bit down, and pixel number i goes to lane i mod 64 -- the row it sits in is found by a binary search over the rows'
    // running counts (ds_bpermute, as fx_noncorner_many) -- and a lane evaluates just the components its pixel needs.
    const uint32_t lane = (uint32_t)__lane_id();
    const bool have = lane < nrows;
    const float c0 = 0x1.6a09e6p-1f;
    const uint32_t keys = have ? q[lane * ROWQ_WORDS + 12] : 0xFFFFFFFFu;
    const uint32_t todo = ~(keys & (keys >> 8) & (keys >> 16) & (keys >> 24)) & 0xFFu;   // bit 7 - i: pixel column i of the row is marked
    const uint32_t cnt = (uint32_t)__popc(todo), incl = wave_scan_incl(cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    for (uint32_t base = 0; base < total; base += 64) {   // wave-uniform: once, unless more than 64 pixels have gathered
        const uint32_t i = base + lane;
        const bool act = i < total;
        uint32_t owner = 0;   // rows whose running count is <= i
#pragma unroll
        for (uint32_t s = 32; s; s >>= 1) {
            const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((owner + s - 1) << 2), (int)incl);
            owner += v <= i ? s : 0u;
        }
        owner = act ? owner : 63u;
        uint32_t kth = i - (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)(incl - cnt));
        uint32_t bits = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)todo);
        const uint32_t okeys = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)keys);
        kth = act ? kth : 0u;
        while (__ballot(kth != 0)) {   // wave-uniform: at most seven rounds of three instructions
            if (kth) {
                bits &= ~(1u << (31u - (uint32_t)__builtin_clz(bits)));
                --kth;
            }
        }
        if (!(act && bits)) continue;   // (lanes without a pixel sit the rest out; nothing below is wave-wide)
        const uint32_t bit = 31u - (uint32_t)__builtin_clz(bits), y = 7u - bit;
        lds_cu32* const e = q + owner * ROWQ_WORDS;
        const uint32_t where = e[13];
        // the pixel's rounded fast samples
        const __attribute__((address_space(3))) uint16_t* const eh = reinterpret_cast<const __attribute__((address_space(3))) uint16_t*>(e);
        int S[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) S[c] = (int)(float)__builtin_bit_cast(_Float16, eh[c * 8 + y]) + 128;
        const uint32_t x = where & 7u;   // pixel row
        lds_cf64* const cx = s_cos + x * 8;
        lds_cf64* const cy = s_cos + y * 8;
        // the components the entry does not vouch for: MCU::computeIDCT's sum (MCU.cpp:184-198) for a block that has nothing outside
        // (0,0), (0,1), (1,0), (1,1), in its order; a zero term leaves the float accumulator as it is (x + (+-0) == x), so none needs a
        // test.  cos((2x+1) 0 pi/16) == 1.0 exactly.  (fx_noncorner_* have settled what sits in other blocks.)
        uint32_t need = (((okeys >> bit) & 1u) ? 0u : 1u) | (((okeys >> (8 + bit)) & 1u) ? 0u : 2u) | (((okeys >> (16 + bit)) & 1u) ? 0u : 4u);
        while (need) {   // per lane: nearly always one component
            const uint32_t c = (uint32_t)__builtin_ctz(need);
            need &= need - 1u;
            lds_cu32* const qi = s_qi + (c ? 64 : 0);
            const uint32_t wa = e[14 + 2 * c], wb = e[15 + 2 * c];
            const float fc00 = (c0 * c0) * (float)((int)(short)(wa & 0xFFFF) * (int)qi[0]), fc01 = (c0 * 1.0f) * (float)(((int)wa >> 16) * (int)qi[1]),
                        fc10 = (1.0f * c0) * (float)((int)(short)(wb & 0xFFFF) * (int)qi[8]), fc11 = (float)(((int)wb >> 16) * (int)qi[9]);
            float sum = fc00;
            sum = (float)((double)sum + (double)fc01 * cy[1]);
            sum = (float)((double)sum + (double)fc10 * cx[1]);
            sum = (float)((double)sum + ((double)fc11 * cx[1]) * cy[1]);
            const int v = level_shift((float)(0.25 * (double)sum));
            S[0] = c == 0 ? v : S[0], S[1] = c == 1 ? v : S[1], S[2] = c == 2 ? v : S[2];
        }
        const uint32_t px = colour_exact(S[0], S[1], S[2]);
        // the pixel's bytes in the picture
        const uint32_t tile = min(where >> 6, a.ntiles - 1), g = (where >> 3) & 7u;   // MCU of the tile
        const uint32_t trow = a.tiles_w == 1 ? tile : (__umulhi(tile, a.tiles_w_magic) >> a.tiles_w_shift), tcol = tile - trow * a.tiles_w;
        size_t off;
        if (a.rgb_table) {
            const uint32_t img = trow / a.rows_per_img;
            off = (size_t)(reinterpret_cast<uintptr_t>(a.rgb_table[img]) - reinterpret_cast<uintptr_t>(a.rgb)) + (size_t)(trow - img * a.rows_per_img) * 8 * a.pitch;
        } else {
            off = (size_t)trow * 8 * a.pitch;
        }
        uint8_t* const o = a.rgb + off + (size_t)x * a.pitch + (size_t)(tcol * TILE_MCUS + g) * 24 + y * 3;
        o[0] = (uint8_t)px;
        o[1] = (uint8_t)(px >> 8);
        o[2] = (uint8_t)(px >> 16);
    }
}

#ifdef KPEG_K4_STAMP
__device__ unsigned long long g_k4_stamp[8192 * 8];
#endif
// One workgroup of K4_WAVES wavefronts per CU; every wavefront is an independent worker on tiles of 8 MCUs (64 x 8
// pixels), no barrier after the start-up.  The workgroup owns a fixed share of the tiles (chunks dealt round-robin, see
// KPEG_K4_CHUNK) and hands them out through a counter in LDS: on a SIMD the oldest wavefront gets the issue slots first (age
// arbitration), so with a static split the wavefronts of one SIMD finished one after the other -- first 38 us, last 54-66 us,
// the SIMD two-thirds idle at the end (profiles/r02: per-wavefront stamps) -- whereas wavefronts that take tiles as they go
// finish within a tile's time of each other.
//
// Pixels whose fast value cannot be trusted (within the block's bound of a rounding boundary, or a G term too close to an
// integer) are only MARKED in the pixel loop: every lane shifts the signs of its pixels' four keys into four bytes (one
// v_alignbit each, in place of the ANDs that used to merge them: no test, no branch) and packs the rounded samples as f16
// pairs.  Behind the loop ONE wave-uniform test: a tile that has marked pixels (two in five) appends the rows that hold them to
// the wavefront's queue in LDS, 80 bytes a row, at ranks counted from the test's own ballot; samples of blocks that are not
// corner-only are settled there and then, while the tile's blocks stand (fx_noncorner_*); fx_flush settles a full queue.
// Rounds 1 and 2 tested every group of two or four pixel columns where they were made and queued single pixels from inside
// the loop: a scalar branch on a vector compare per group, the ballots and the pushes were 10-13 of the kernel's 69 us
// (profiles/r02_h).  What the pieces cost now (8K workload, profiles/r03_d): the tile loop with its marks
// and the queue 55.6 us, fx_flush 6, the non-corner samples 6 -- a wavefront spends 1.1 + 1.6 us of its 50 in them, but the
// kernel is bound by its instruction stream, and what one wavefront issues the other three of its SIMD wait for.
#ifdef KPEG_K4_WPE
#define KPEG_K4_OCC __attribute__((amdgpu_waves_per_eu(KPEG_K4_WPE, KPEG_K4_WPE)))   // experiments: several smaller workgroups per CU
#else
#define KPEG_K4_OCC
#endif
template <bool COMPACT>
__global__ __launch_bounds__(K4_THREADS) KPEG_K4_OCC void k_idct_colour_fast(IdctParams p, QTables qt)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_tile_all[K4_WAVES][8 * TILE_ROW_STRIDE];
    __shared__ __attribute__((aligned(16))) uint32_t s_img_all[K4_WAVES][IMG_BYTES / 4];   // the tile's 24 blocks, natural order, int16 (compact stream: rebuilt from the records; dense layout: the rows as loaded)
    __shared__ __attribute__((aligned(16))) float s_m[2][64];     // AC input scales, natural order
    __shared__ uint32_t s_next;       // next tile of this workgroup's range to hand out
    __shared__ __attribute__((aligned(16))) uint32_t s_rowq_all[K4_WAVES][ROWQ_CAP * ROWQ_WORDS];   // the wavefronts' queues of pixel rows that have a marked pixel
    __shared__ __attribute__((aligned(16))) uint32_t s_qi[2 * 64];   // quantisers, natural order
    __shared__ double s_cos[64];
    __shared__ uint32_t s_wg[2];      // [0] wavefronts of this workgroup that are done, [1] pixels they settled

#ifdef KPEG_K4_STAMP
    // diagnostic build only (tools/k4_clock.py): the shader clock this kernel runs at = d(s_memtime) / d(s_memrealtime) x 100 MHz
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_flush = 0, stamp_nc = 0, stamp_nflush = 0, stamp_nnc = 0;   // shader cycles in fx_flush / the non-corner block, and how often
#define KPEG_STAMP_BEGIN const unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#define KPEG_STAMP_END(acc, n) acc += __builtin_amdgcn_s_memtime() - stamp_t, n += 1;
#else
#define KPEG_STAMP_BEGIN
#define KPEG_STAMP_END(acc, n)
#endif
    const int tid = threadIdx.x & 63;   // lane of the wavefront
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint8_t* const s_tile = s_tile_all[wave];
    uint32_t* const s_img = s_img_all[wave];
    const int lane8 = tid & 7;          // lane within the MCU group = output pixel row
    const int grp = tid >> 3;           // MCU within the tile, 0..7
    const int u = lane8 < 4 ? 2 * lane8 : 2 * (lane8 - 4) + 1;  // coefficient row this lane loads

    if (threadIdx.x < 128) {
        const int t = threadIdx.x >> 6, k = threadIdx.x & 63;
        s_m[t][k] = 0.25f * cc_of(k >> 3, k & 7) * (float)qt.q[t][k];
        s_qi[t * 64 + k] = qt.q[t][k];
    } else if (threadIdx.x == 128) {
        s_next = 0;
        s_wg[0] = 0;
        s_wg[1] = 0;
    } else if (threadIdx.x >= 192 && threadIdx.x < 256) {
        s_cos[tid] = c_cos[tid];
    }
    // This workgroup's tiles: chunks of KPEG_K4_CHUNK consecutive tiles, dealt round-robin to the workgroups -- tile number k of the
    // workgroup is tile ((k / CHUNK) * workgroups + blockIdx) * CHUNK + k % CHUNK.  A contiguous range per workgroup (rounds 1 and 2)
    // leaves the workgroups that own a patch of marked pixels (they come in clusters: a textured region, a ramp whose blocks tie
    // on their diagonals) working after the others are done; dealt out, a patch is everybody's.
#ifndef KPEG_K4_CHUNK
#define KPEG_K4_CHUNK 8
#endif
    constexpr uint32_t CHUNK = KPEG_K4_CHUNK;
    uint32_t wg_ntiles;
    {
        const uint32_t round = gridDim.x * CHUNK, full = p.ntiles / round, rem = p.ntiles - full * round;
        const uint32_t mine = rem > blockIdx.x * CHUNK ? min(rem - blockIdx.x * CHUNK, CHUNK) : 0u;
        wg_ntiles = full * CHUNK + mine;
    }
    auto tile_of = [&](uint32_t k) -> uint32_t { return ((k / CHUNK) * gridDim.x + blockIdx.x) * CHUNK + k % CHUNK; };
    LaneConst lc;
    lc.q0[0] = (float)qt.q[0][u * 8];
    lc.q0[1] = (float)qt.q[1][u * 8];
    lc.cc0 = cc_of(u, 0);
    // Column-pass constants of this lane: rows 0,2,4,6 live on lanes 0..3, lane x accumulates E_x with cos((2x+1) 2k pi/16);
    // rows 1,3,5,7 on lanes 4..7, lane 7-x accumulates -O_x with -cos((2x+1)(2k+1) pi/16).  Literals picked by a select
    // chain: a table in memory indexed by the lane cost every wavefront sixteen serialised loads before its first tile.
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float r = cosf_tab(1 * (2 * k));
#pragma unroll
        for (int l = 1; l < 8; ++l) {
            const float c = l < 4 ? cosf_tab((2 * l + 1) * (2 * k)) : -cosf_tab((2 * (7 - l) + 1) * (2 * k + 1));
            r = lane8 == l ? c : r;
        }
        lc.k[k] = r;
    }
    // combine: out = own + mirror * s.  Even lane x: E_x - (-O_x) -> s = -1;
    // odd lane (pixel row 7-x): E_x - O_x = mirror(E_x) + own(-O_x) -> s = +1.
    lc.s = lane8 < 4 ? -1.0f : 1.0f;
    __syncthreads();   // the tables and the tile counter stand (the only workgroup barrier)
    uint32_t fx_lane = 0;      // marked pixels of this lane so far
    uint32_t* const s_rowq = s_rowq_all[wave];
    uint32_t qcount = 0;       // rows in the queue (wave-uniform)
    auto flush_rows = [&]() {
        KPEG_STAMP_BEGIN
        // the queued rows' tiles have been handed to the memory system (write_back); their stores must have been performed before bytes
        // of theirs are patched
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const FxArgs fa = {p.rgb, p.rgb_table, p.pitch, p.tiles_w, p.tiles_w_magic, p.tiles_w_shift, p.ntiles, p.rows_per_img};
#ifndef KPEG_FX_NO_FLUSH
        fx_flush(fa, (lds_cu32*)s_rowq, qcount, (lds_cu32*)s_qi, (lds_cf64*)s_cos);
#endif
        qcount = 0;
        KPEG_STAMP_END(stamp_flush, stamp_nflush)
    };
    // takes the next tile of the workgroup's range: the value is asked for one tile ahead, so the LDS round trip is not waited for
    auto take_tile = [&]() -> uint32_t {
        uint32_t t = 0;
        if (tid == 0) t = atomicAdd(&s_next, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };

    // Coalesced write-back of a finished tile from LDS: 8 rows x nm*24 bytes as 16-byte chunks
    // (8 x 12 = 96 chunks: lanes 0..47 store chunk k of row r and of row r + 4; two registers per lane hold it all).
    // It is issued one iteration late, ahead of the next loads, so that waiting for a tile's
    // coefficients never waits for the stores that follow them in issue order.
    // (the lane's two offsets are worked out anew for every tile, from a lane number the compiler cannot see through: kept across the
    // loop they are two more registers than there are, and the spilled one came back through a scratch load whose wait was a wait for
    // the next tile's coefficients -- 20 us of K4, profiles/r03_d)
    const bool pitch16 = ((reinterpret_cast<uintptr_t>(p.rgb) | p.pitch) & 15) == 0;
    // Where a tile's pixels go, as an offset from the kernel argument p.rgb -- also in table mode: a pointer
    // loaded from memory has no known address space, its stores would be flat_store, and LDS waits wait for
    // those.  Computed when the tile is taken (the table load then travels with the coefficient loads), used
    // one iteration later by write_back: a load inside write_back would make it wait for the next tile's
    // coefficients.
    auto tile_offset = [&](uint32_t trow, uint32_t m0) -> size_t {
        if (p.rgb_table) {   // wave-uniform
            const uint32_t img = trow / p.rows_per_img;
            // scalar load (the index is wave-uniform): a vector load here would sit behind the coefficient loads in
            // vmcnt order, and the wait for it would be a wait for them
            unsigned long long ptr;
            asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                         : "=s"(ptr)
                         : "s"(p.rgb_table), "s"(img * 8u)
                         : "memory");
            return (size_t)(ptr - reinterpret_cast<uintptr_t>(p.rgb)) + (size_t)(trow - img * p.rows_per_img) * 8 * p.pitch + (size_t)m0 * 24;
        }
        return (size_t)trow * 8 * p.pitch + (size_t)m0 * 24;
    };
    auto write_back = [&](size_t off, uint32_t nm) {
        // the lane's share of a tile stays a 32-bit offset added to a scalar base (global_store with an SGPR base):
        // left alone, the compiler adds p.rgb to every lane offset ahead of the loop and keeps 64-bit addresses in VGPRs
        uint32_t wl = (uint32_t)tid;
        asm volatile("" : "+v"(wl));
        const uint32_t wb_r = wl / 12u, wb_k = wl - wb_r * 12u;
        const uint32_t wb_lds = wb_r * TILE_ROW_STRIDE + wb_k * 16, oA = wb_r * p.pitch + wb_k * 16;
        uint8_t* base = p.rgb + off;
        if (nm == TILE_MCUS && pitch16) {
            if (tid < 48) {
#ifdef KPEG_ABLATE_STORES
                if (p.ntiles == 1) *reinterpret_cast<uint4*>(base + oA) = *reinterpret_cast<const uint4*>(s_tile + wb_lds);
#else
                // nontemporal: the pixels are not read again here, and written the ordinary way they push the
                // coefficients K1/K2 have just produced out of the Infinity Cache ahead of this kernel's own loads
                // (K4 inside the decode: 0.084 -> 0.073 ms)
                __builtin_nontemporal_store(*reinterpret_cast<const uint4v*>(s_tile + wb_lds), reinterpret_cast<uint4v*>(base + oA));
                __builtin_nontemporal_store(*reinterpret_cast<const uint4v*>(s_tile + wb_lds + 4 * TILE_ROW_STRIDE), reinterpret_cast<uint4v*>(base + (size_t)4 * p.pitch + oA));
#endif
            }
        } else {
            const uint32_t per_row = nm * 6;  // 4-byte pieces
            for (uint32_t c = tid; c < 8 * per_row; c += 64) {
                const uint32_t r = c / per_row, k = c - r * per_row;
                *reinterpret_cast<uint32_t*>(base + (size_t)r * p.pitch + k * 4) =
                    *reinterpret_cast<const uint32_t*>(s_tile + r * TILE_ROW_STRIDE + k * 4);
            }
        }
    };

    bool have_prev = false;
    size_t prev_off = 0;
    uint32_t prev_nm = 0;
    // A tile's inputs: one 16-byte coefficient row of each component block and the three blocks' bounds.  They are asked
    // for one tile ahead (a second register set), so that a wavefront's tile costs it its instructions and not a memory
    // round trip on top -- with four or five wavefronts per SIMD the others cannot cover that wait.
    // Compact stream: the tile's records (4 bytes per non-zero AC coefficient, ~90 per tile on the 8K workload against 3 KiB of
    // dense rows), its 24 DC values and the bounds are asked for one tile ahead, the two words of the first-record table
    // two tiles ahead (they say where the records are); the blocks are rebuilt in LDS when the tile's turn comes.
    struct TileIn {
        uint4 d0, d1, d2;        // dense layout: this lane's coefficient rows
        float e0, e1, e2;
        uint32_t r0, r1, dcw;    // compact stream: records rs + lane, rs + 64 + lane; DC of block `lane` of the tile
        uint32_t rs, rn;         // ... first record and record count (wave-uniform)
    };
    auto tile_records = [&](uint32_t tk, uint32_t& rs, uint32_t& rn) {
        const uint32_t tile = tile_of(tk);
        const uint32_t a = p.tile_start[tile], b = p.tile_start[tile + 1];   // wave-uniform addresses: scalar loads
        rs = a;
        rn = (b >= a && b <= p.rec_cap && b - a <= 24u * 63u) ? b - a : 0u;   // (a corrupt stream may leave anything in the table)
    };
    auto issue_loads = [&](uint32_t tk, TileIn& in, uint32_t rs, uint32_t rn) {
        const uint32_t tile = tile_of(tk);
        const uint32_t trow = tile_row(p, tile), tcol = tile - trow * p.tiles_w;
        const uint32_t m0 = tcol * TILE_MCUS;
        const uint32_t nm = min((uint32_t)TILE_MCUS, p.mcus_w - m0);
        // MCU of this lane's group (groups beyond the image re-read the tile's first MCU; their pixels are never stored).
        // Addresses are a scalar base (the tile's first MCU) plus a 32-bit lane offset: global_load with an SGPR base.
        uint32_t lane_mcu = (uint32_t)grp < nm ? (uint32_t)grp : 0u;
        asm volatile("" : "+v"(lane_mcu));
#ifdef KPEG_ABLATE_SAMETILE
        // timing experiment: every tile reads the coefficients of one of 64 tiles (cache-resident): what the kernel costs without its HBM reads
        const size_t mcu_ld = (size_t)(tile & 63u) * TILE_MCUS;
#else
        const size_t mcu_ld = (size_t)trow * p.mcus_w + m0;
#endif
        if constexpr (COMPACT) {
            in.rs = rs, in.rn = rn;
            const uint32_t* rp = p.rec + rs;
            uint32_t l = (uint32_t)tid;
            asm volatile("" : "+v"(l));
            in.r0 = l < rn ? rp[l] : 31u;            // (block 31 does not exist: skipped; nontemporal loads here: K4 63.4 -> 66.4 us)
            in.r1 = l + 64u < rn ? rp[l + 64u] : 31u;
            in.dcw = l < 24u ? (uint32_t)(uint16_t)p.dc16[mcu_ld * 3 + l] : 0u;
        } else {
            const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(p.coef + mcu_ld * 192) + (lane_mcu * 384u + (uint32_t)u * 16u));
            in.d0 = src[0], in.d1 = src[8], in.d2 = src[16];
        }
        const float* eb = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(p.ebound + mcu_ld * 3) + lane_mcu * 12u);
        in.e0 = eb[0], in.e1 = eb[1], in.e2 = eb[2];
    };
    // The dense layout's inputs (12 VGPRs of rows) are NOT asked for ahead: measured no gain (the kernel is bound by HBM
    // bytes and by its own instruction stream, not by that round trip), and the registers are needed elsewhere.
    constexpr bool PREFETCH = COMPACT;
    uint32_t tilek_cur = take_tile(), tilek_next = take_tile();
    TileIn cur, nxt;
    uint32_t rs_next = 0, rn_next = 0, rs_after = 0, rn_after = 0;
    if (PREFETCH && tilek_cur < wg_ntiles) {
        uint32_t rs = 0, rn = 0;
        if constexpr (COMPACT) tile_records(tilek_cur, rs, rn);
        issue_loads(tilek_cur, cur, rs, rn);
    }
    if constexpr (COMPACT)
        if (tilek_next < wg_ntiles) tile_records(tilek_next, rs_next, rn_next);
    for (;;) {
        // The tile's memory phase (hand-out, rebuild of the blocks in LDS, the next tile's loads, the previous tile's write-back) is issued
        // ahead of the other wavefronts' arithmetic: whoever has latencies to start starts them first.  0.8 us of K4 (profiles/r03_m_k4_wave_priority_ab.txt;
        // the other way round -- the arithmetic first -- costs 3).
#ifndef KPEG_K4_PRIO_MEM
#define KPEG_K4_PRIO_MEM 2
#define KPEG_K4_PRIO_ALU 0
#endif
        __builtin_amdgcn_s_setprio(KPEG_K4_PRIO_MEM);
        const uint32_t tilek = tilek_cur;              // this tile's number inside the workgroup's range
        if (!(tilek < wg_ntiles)) break;               // wave-uniform
        const uint32_t tile = tile_of(tilek);
        const uint32_t tilek_after = take_tile();
        if constexpr (COMPACT)
            if (tilek_after < wg_ntiles) tile_records(tilek_after, rs_after, rn_after);   // two tiles ahead: nothing waits for these
        if constexpr (!PREFETCH) issue_loads(tilek, cur, 0, 0);
        const uint32_t trow = tile_row(p, tile), tcol = tile - trow * p.tiles_w;
        const uint32_t m0 = tcol * TILE_MCUS;                       // first MCU column of the tile
        const uint32_t nm = min((uint32_t)TILE_MCUS, p.mcus_w - m0);  // MCUs in this tile
        const bool active = (uint32_t)grp < nm;
        uint4 d0, d1, d2;
        if constexpr (COMPACT) {
            // rebuild the tile's 24 blocks in LDS (zero, scatter the records, the DC values), then read this lane's rows
            uint4v* img4 = reinterpret_cast<uint4v*>(s_img);
            const uint4v z = {0u, 0u, 0u, 0u};
            img4[tid] = z, img4[tid + 64] = z, img4[tid + 128] = z;
            uint16_t* img16 = reinterpret_cast<uint16_t*>(s_img);
            auto put = [&](uint32_t r) {
                const uint32_t bm = r & 31u;
                if (bm < 24u) img16[bm * 64u + ((r >> 8) & 63u)] = (uint16_t)(r >> 16);
            };
            put(cur.r0);
            put(cur.r1);
            for (uint32_t b0 = 128; b0 < cur.rn; b0 += 64) {   // (dense tiles only)
                const uint32_t idx = b0 + (uint32_t)tid;
                if (idx < cur.rn) put(p.rec[cur.rs + idx]);
            }
            if (tid < 24) img16[tid * 64] = (uint16_t)cur.dcw;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4* rows = reinterpret_cast<const uint4*>(s_img) + (grp * 3) * 8 + u;
            d0 = rows[0], d1 = rows[8], d2 = rows[16];
        } else {
            d0 = cur.d0, d1 = cur.d1, d2 = cur.d2;
            // the rows as loaded also go to the tile's image in LDS (three 16-byte stores per lane): what settles the marked pixels
            // reads the blocks there, as it does with the compact stream (round 3's first build read them again from memory, one lane
            // per pixel row: dense content -- noise, photographs at high quality -- took K4 10-30 % longer than round 2's)
            uint4* rows = reinterpret_cast<uint4*>(s_img) + (grp * 3) * 8 + u;
            rows[0] = d0, rows[8] = d1, rows[16] = d2;
        }
        const float e0 = cur.e0, e1 = cur.e1, e2 = cur.e2;
        if constexpr (PREFETCH) {
            // the next tile's inputs travel while this one is computed (asked for here, behind the rebuild: this tile's
            // records are dead by now and their registers free)
            if (tilek_next < wg_ntiles) issue_loads(tilek_next, nxt, rs_next, rn_next);
            asm volatile("" ::: "memory");
        }
        const size_t cur_off = tile_offset(trow, m0);
        if (have_prev) write_back(prev_off, prev_nm);  // LDS still holds the previous tile
        have_prev = true;
        prev_off = cur_off;
        prev_nm = nm;

        // Quantised high frequencies are mostly zero: if no block of this wavefront has a coefficient outside
        // its top-left 6x6 (luma) / 4x4 (chroma) corner, the terms of the empty rows and columns are left out
        // (same floats as the full transform, see row_idct8).  8K q75: luma 6x6 for 98 % of the tiles, chroma 4x4 for 99 %.
        // All four wave-uniform conditions of a tile are taken together, ahead of the first branch on any of them: a scalar
        // branch on a vector compare stalls the wavefront until the compare has left the vector pipe, once instead of four times.
        __builtin_amdgcn_s_setprio(KPEG_K4_PRIO_ALU);
        const bool big0 = __ballot((d0.w | (u >= 6 ? (d0.x | d0.y | d0.z) : 0u)) != 0) != 0;
        const bool big1 = __ballot(((d1.z | d1.w) | (u >= 4 ? (d1.x | d1.y) : 0u)) != 0) != 0;
        const bool big2 = __ballot(((d2.z | d2.w) | (u >= 4 ? (d2.x | d2.y) : 0u)) != 0) != 0;
        // chroma samples of this MCU may exceed the f32 colour arithmetic's proven range (see block_ebound)
        const bool wide = (((__float_as_uint(e1) | __float_as_uint(e2)) & 1u) != 0) && active;
        const bool any_wide = __ballot(wide) != 0;   // wave-uniform, rare

        float v[3][8];
#ifdef KPEG_ABLATE_IDCT
        for (int i = 0; i < 8; ++i) {
            v[0][i] = __uint_as_float(d0.x + i) * 1e-30f;
            v[1][i] = __uint_as_float(d1.y + i) * 1e-30f;
            v[2][i] = __uint_as_float(d2.z + i) * 1e-30f;
        }
        (void)big0, (void)big1, (void)big2;
#else
        if (big0) block_fast<8>(d0, lc, &s_m[0][u * 8], 0, v[0]);
        else block_fast<6>(d0, lc, &s_m[0][u * 8], 0, v[0]);
        if (big1) block_fast<8>(d1, lc, &s_m[1][u * 8], 1, v[1]);
        else block_fast<4>(d1, lc, &s_m[1][u * 8], 1, v[1]);
        if (big2) block_fast<8>(d2, lc, &s_m[1][u * 8], 1, v[2]);
        else block_fast<4>(d2, lc, &s_m[1][u * 8], 1, v[2]);
#endif
        // |fast - rint(fast)| + nthr >= 0  <=>  within the block's bound of a rounding boundary
#if !KPEG_K4_MASK_MARKS
        const float nthr0 = fabsf(e0) - 0.5f, nthr1 = fabsf(e1) - 0.5f, nthr2 = fabsf(e2) - 0.5f;
#endif

        // Level shift + colour for the 8 pixels of this lane's row.  Per pixel the sign of one word says whether the
        // reference-order evaluation is needed (sign clear): a fast value within its block's bound of a rounding boundary,
        // or a G term too close to an integer for the f32 arithmetic.  Nearly every wavefront has a few such pixels (true
        // ties are structural: equal and opposite (0,1)/(1,0) terms cancel on a block's diagonal and leave DC/8 = n + 0.5
        // exactly); their signs are shifted into `ub`, one bit per pixel column.
        uint32_t pk[6] = {0, 0, 0, 0, 0, 0};
        // KPEG_K4_MASK_MARKS=1 (an experiment, measured and not the default): the pixel loop as planned below is 23 instead of 30 issue
        // cycles of marking per pixel, but its ~24 scalar registers more push the kernel from 6 to 52 scalar spills (v_writelane /
        // v_readlane around every tile): K4 63.2 -> 70.6 us (profiles/r03_k_k4_mask_marks_ab.txt).  It needs the kernel's scalar
        // live set cut first (its arguments re-read per tile instead of kept).
#ifndef KPEG_K4_MASK_MARKS
#define KPEG_K4_MASK_MARKS 0
#endif
#if KPEG_K4_MASK_MARKS
        // The marks as lane masks in scalar registers: every key is ONE vector compare (|v - rint v| against the block's threshold,
        // written straight to a scalar register pair) in place of an add and a v_alignbit; the scalar unit ORs a pixel column's four
        // masks, and one v_addc shifts the column's bit into the lane's byte (carry-in = the mask).  Which COMPONENT a mark is for is
        // kept per pixel row only (four more masks): fx_flush evaluates a marked pixel's component in the reference's order when the
        // row has a mark of that component at all (rows have 1.03 marked pixels on average; evaluating more than needed is always right).
        const float thr0 = 0.5f - fabsf(e0), thr1 = 0.5f - fabsf(e1), thr2 = 0.5f - fabsf(e2);   // (== -nthr: the same decisions as the sign of |d| + nthr)
        unsigned long long rowY = 0, rowB = 0, rowR = 0, rowG = 0;   // lanes whose pixel row has a mark of the component / of the G term
        uint32_t ubM = 0;                                            // bit 7 - i = pixel column i is MARKED
        const unsigned long long wide_mask = __ballot(wide);
#else
        uint32_t ubY = 0, ubB = 0, ubR = 0, ubG = 0;   // per component and for the G term: bit 7 - i = pixel column i is safe
#endif
        uint32_t hp[3][4];                              // the rounded samples as f16 pairs
        float py = 0.0f, pb = 0.0f, pr = 0.0f;          // the even column's, until its odd neighbour's are there
        // The loop exists twice: with the in-lane double colour conversion of `wide` MCUs and (nearly always) without.
        auto pixel_loop = [&](auto with_wide) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float vy = v[0][i], vb = v[1][i], vr = v[2][i];
                const float ry = __builtin_rintf(vy), rb = __builtin_rintf(vb), rr = __builtin_rintf(vr);
#if !KPEG_K4_MASK_MARKS
                // >= 0: the fast value is within its block's bound of a rounding boundary
                const float fy = fabsf(vy - ry) + nthr0, fb = fabsf(vb - rb) + nthr1, fr = fabsf(vr - rr) + nthr2;
#endif
                // colour from the three rounded samples (minus the level shift); dt = how far the G term's t is from
                // an integer, as seen by the f32 arithmetic
                // v_cvt_pk_u8_f32 rounds to nearest-even and saturates.  All three channels are handed to it 0.499 below
                // their value: floor(Y + k c) for R and B -- the fractions of 1.402 c and 1.772 c are multiples of 1/500 and
                // 1/250, so value - 0.499 lies in (n - 0.5, n + 0.5) with 0.001 to spare on either side, more than the f32
                // roundings of the two operations can move it for |Y| <= 4100, |c| <= 249 -- and the integer Y - ceil(t) for G
                // (tests/test_tables.py checks every case).  No floor, no separate level-shift add.
                const float yo = ry + 127.501f;
                float R = __builtin_fmaf(rr, 1.402f, yo);
                float B = __builtin_fmaf(rb, 1.772f, yo);
                const float t = __builtin_fmaf(rr, 0.714136f, rb * 0.344136f);
                const float tc = ceilf(t);
                float G = yo - tc;
                // G needs the reference's double arithmetic where t is within KPEG_G_DELTA of a non-zero integer
                // (t == 0, i.e. Cb = Cr = 128, is exact in the reference too; non-zero |t| is >= 8e-6, so
                // |t| * 2^17 >= 1 there).  Both conditions and the three sample keys are combined as sign bits with
                // and/or (1.7-cycle instructions; max/max3/rndne issue at 2.7): the sign of `safe` is set iff nothing
                // about this pixel is unsafe.
#if KPEG_K4_MASK_MARKS
                // (not-less-than: a NaN marks)
                const unsigned long long mY = __ballot(!(fabsf(vy - ry) < thr0)), mB = __ballot(!(fabsf(vb - rb) < thr1)), mR = __ballot(!(fabsf(vr - rr) < thr2));
                unsigned long long mG = __ballot(!(fabsf((tc - t) - 0.5f) < (0.5f - KPEG_G_DELTA))) & __ballot(!(fabsf(t) < 4.0e-6f));   // within DELTA of an integer, and t != 0
                uint32_t kg = 0;
#else
                const float ka = fabsf((tc - t) - 0.5f) - (0.5f - KPEG_G_DELTA);                // >= 0: within DELTA of an integer
                const float kb = __builtin_fmaf(fabsf(t), 131072.0f, KPEG_G_DELTA - 1.0f);      // >= 0: t != 0
                uint32_t kg = __float_as_uint(ka) | __float_as_uint(kb);                         // sign clear: G is unsafe
#ifdef KPEG_ABLATE_GKEY
                kg = 0x80000000u;   // timing experiment: no G key (wrong on 36 chroma pairs)
#endif
#endif
                if (decltype(with_wide)::value) {
                    if (wide) {
                        // out of the f32 colour arithmetic's range: the reference's own double arithmetic on the rounded samples
                        const uint32_t px = colour_exact((int)ry + 128, (int)rb + 128, (int)rr + 128);
                        R = (float)(px & 0xFF);
                        G = (float)((px >> 8) & 0xFF);
                        B = (float)(px >> 16);
                        kg = 0x80000000u;   // G is exact here
                    }
                }
                pk[(3 * i) >> 2] = pk_u8(R, (3 * i) & 3, pk[(3 * i) >> 2]);
                pk[(3 * i + 1) >> 2] = pk_u8(G, (3 * i + 1) & 3, pk[(3 * i + 1) >> 2]);
                pk[(3 * i + 2) >> 2] = pk_u8(B, (3 * i + 2) & 3, pk[(3 * i + 2) >> 2]);
#if defined(KPEG_ABLATE_PUSH)
#if KPEG_K4_MASK_MARKS
                (void)mY, (void)mB, (void)mR, (void)mG, (void)kg;
#else
                (void)fy, (void)fb, (void)fr, (void)kg;   // timing experiment: no unsafe-pixel arithmetic survives
#endif
#else
#if KPEG_K4_MASK_MARKS
                (void)kg;
                if (decltype(with_wide)::value) mG &= ~wide_mask;   // (their G comes from the reference's own arithmetic)
                rowY |= mY, rowB |= mB, rowR |= mR, rowG |= mG;
                {
                    const unsigned long long mp = (mY | mB) | (mR | mG);
                    unsigned long long carry;
                    asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(ubM), "=&s"(carry) : "s"(mp));   // ubM = ubM << 1 | the lane's bit of mp
                }
#else
                // the signs into the four bytes: u << 1 | sign (one v_alignbit each, in place of the ANDs that used to merge them)
                ubY = __builtin_amdgcn_alignbit(ubY, __float_as_uint(fy), 31);
                ubB = __builtin_amdgcn_alignbit(ubB, __float_as_uint(fb), 31);
                ubR = __builtin_amdgcn_alignbit(ubR, __float_as_uint(fr), 31);
                ubG = __builtin_amdgcn_alignbit(ubG, kg, 31);
#endif
                if (i & 1) {
                    // two columns' rounded samples as an f16 pair (exact: integers, |.| <= 2048 wherever the bound is finite enough to matter;
                    // fx_flush treats a larger one as unsafe)
                    hp[0][i >> 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(py, ry));
                    hp[1][i >> 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(pb, rb));
                    hp[2][i >> 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(pr, rr));
                } else {
                    py = ry, pb = rb, pr = rr;
                }
#endif
            }
        };
        if (any_wide) pixel_loop(std::true_type{});
        else pixel_loop(std::false_type{});
        {
            // 24 bytes of pixel row lane8, MCU grp (groups beyond nm write garbage that is never stored)
            uint2* dst = reinterpret_cast<uint2*>(s_tile + lane8 * TILE_ROW_STRIDE + grp * 24);
            dst[0] = make_uint2(pk[0], pk[1]);
            dst[1] = make_uint2(pk[2], pk[3]);
            dst[2] = make_uint2(pk[4], pk[5]);
        }
#if !defined(KPEG_ABLATE_PUSH)
        {
            // One compare, one ballot: the tile loop's only test on what the pixel loop found.  The rows that have a marked pixel join the
            // queue, at ranks counted from the same ballot; the count is a scalar.
#if KPEG_K4_MASK_MARKS
            // the rows that have a mark: a scalar OR of the four masks, no vector compare and no wait for one
            const unsigned long long mbal = ((rowY | rowB) | (rowR | rowG)) & __ballot(active);
#else
            const uint32_t marked = active ? ~(ubY & ubB & ubR & ubG) & 0xFFu : 0u;   // bit 7 - i = pixel column i needs the reference-order evaluation
            const unsigned long long mbal = __ballot(marked != 0);
#endif
            if (mbal) {   // wave-uniform: two tiles in five on the 8K workload
#if KPEG_K4_MASK_MARKS
                const uint32_t marked = ((mbal >> tid) & 1ull) ? ubM & 0xFFu : 0u;   // bit 7 - i = pixel column i needs the reference-order evaluation
                // per component: bit 7 - i = pixel column i is vouched for -- every unmarked one, and the marked ones too unless the row has a mark of the component
                const uint32_t ubY = ((rowY >> tid) & 1ull) ? ~marked : 0xFFu, ubB = ((rowB >> tid) & 1ull) ? ~marked : 0xFFu,
                               ubR = ((rowR >> tid) & 1ull) ? ~marked : 0xFFu, ubG = ((rowG >> tid) & 1ull) ? ~marked & 0xFFu : 0xFFu;
#endif
#if !defined(KPEG_COUNT_NC) && !defined(KPEG_COUNT_NCTILES) && !defined(KPEG_COUNT_ROWS)
                fx_lane += (uint32_t)__popc(marked);
#elif defined(KPEG_COUNT_ROWS)
                fx_lane += marked ? 1u : 0u;
#endif
                if (!p.skip_exact) {
                    const uint32_t nrow = (uint32_t)__popcll(mbal);
                    if (qcount + nrow > (uint32_t)ROWQ_CAP) flush_rows();   // (the rows queued so far are of tiles already written back)
                    const uint32_t slot = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(mbal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mbal, 0));
                    uint32_t* const entry = s_rowq + (marked ? slot : 0u) * ROWQ_WORDS;
                    uint32_t nc = 0, inf = 0;
                    if (marked) {
                        uint4v* row = reinterpret_cast<uint4v*>(entry);
                        const uint4v h0 = {hp[0][0], hp[0][1], hp[0][2], hp[0][3]}, h1 = {hp[1][0], hp[1][1], hp[1][2], hp[1][3]}, h2 = {hp[2][0], hp[2][1], hp[2][2], hp[2][3]};
                        uint4v h3 = {(ubY & 0xFFu) | ((ubB & 0xFFu) << 8) | ((ubR & 0xFFu) << 16) | (ubG << 24), (tile << 6) | (uint32_t)tid, 0u, 0u};
                        uint4v h4 = {0u, 0u, 0u, 0u};
                        {
                            uint32_t gl = (uint32_t)tid;
                            asm volatile("" : "+v"(gl));   // (worked out here, not kept in a register across the loop)
                            const uint32_t* b32 = s_img + (gl >> 3) * 3 * 32;   // the MCU's three blocks, 32 words each
                            h3.z = b32[0], h3.w = b32[4], h4.x = b32[32], h4.y = b32[36], h4.z = b32[64], h4.w = b32[68];
                            // samples of blocks that are not corner-only (the bound's sign is clear) are settled while the blocks stand
                            nc = (__float_as_int(e0) < 0 ? 0u : ~ubY & 0xFFu) | (__float_as_int(e1) < 0 ? 0u : (~ubB & 0xFFu) << 8) | (__float_as_int(e2) < 0 ? 0u : (~ubR & 0xFFu) << 16);
                            inf = (e0 == __builtin_inff() ? 1u : 0u) | (e1 == __builtin_inff() ? 2u : 0u) | (e2 == __builtin_inff() ? 4u : 0u);
                        }
                        row[0] = h0, row[1] = h1, row[2] = h2, row[3] = h3, row[4] = h4;
                    }
                    qcount += nrow;
                    {
#if defined(KPEG_COUNT_NC)
                        fx_lane += (uint32_t)__popc(nc);
#elif defined(KPEG_COUNT_NCTILES)
                        fx_lane += (tid == 0 && __ballot(nc != 0)) ? 1u : 0u;
#endif
                        if (__ballot(nc != 0)) {   // wave-uniform: one marked tile in ten
                            KPEG_STAMP_BEGIN
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            const uint32_t total = (uint32_t)__popcll(__ballot((nc & 0xFFu) != 0)) + (uint32_t)__popcll(__ballot((nc & 0xFF00u) != 0)) +
                                                   (uint32_t)__popcll(__ballot((nc & 0xFF0000u) != 0));   // rows-and-components
#ifndef KPEG_FX_FEW_MAX
#define KPEG_FX_FEW_MAX 3u
#endif
                            if (total > KPEG_FX_FEW_MAX || __ballot(inf != 0))
#ifdef KPEG_FX_SKIP_MANY
                                ;
                            else if (false)
#endif
                                fx_noncorner_many((lds_u32*)entry, (lds_cu32*)s_img, (lds_cu32*)s_qi, (lds_cf64*)s_cos, nc, inf,
                                                  (__attribute__((address_space(3))) uint8_t*)(s_tile + lane8 * TILE_ROW_STRIDE + grp * 24));
                            else
#ifdef KPEG_FX_SKIP_FEW
                                if (false)
#endif
                                fx_noncorner_few((lds_u32*)entry, (lds_cu32*)s_img, (lds_cu32*)s_qi, (lds_cf64*)s_cos, nc);
                            KPEG_STAMP_END(stamp_nc, stamp_nnc)
                        }
                    }
                }
            }
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the next iteration writes this tile back before its own colour phase overwrites the LDS tile
        if constexpr (PREFETCH) cur = nxt;
        tilek_cur = tilek_next;
        tilek_next = tilek_after;
        rs_next = rs_after, rn_next = rn_after;
    }
    if (have_prev) write_back(prev_off, prev_nm);
#ifdef KPEG_K4_STAMP
    const unsigned long long stamp_r1 = __builtin_amdgcn_s_memrealtime();   // the tile loop is done
#endif
    if (qcount) flush_rows();   // what is left (the last tile has just been handed to the memory system)
#ifdef KPEG_K4_STAMP
    if (tid == 0) {
        const unsigned long long dc = __builtin_amdgcn_s_memtime() - stamp_c0, dr = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        const uint32_t w = blockIdx.x * K4_WAVES + wave;
        if (w < 8192) {
            // every wavefront's start, end (100 MHz ticks), lifetime in shader cycles, to a slot of its own (a buffer nothing else reads)
            g_k4_stamp[32768 + w * 4 + 0] = stamp_r1;
            g_k4_stamp[32768 + w * 4 + 1] = stamp_flush | (stamp_nflush << 48);
            g_k4_stamp[32768 + w * 4 + 2] = stamp_nc | (stamp_nnc << 48);
            g_k4_stamp[32768 + w * 4 + 3] = fx_lane;
            g_k4_stamp[w * 4 + 0] = stamp_r0;
            g_k4_stamp[w * 4 + 1] = stamp_r0 + dr;
            g_k4_stamp[w * 4 + 2] = dc;
            // XCC_ID [35:32] | HW_ID [31:0] (wave [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13])
            g_k4_stamp[w * 4 + 3] = ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15u) << 32) |
                                    __builtin_amdgcn_s_getreg(4 | (31 << 11));
        }
    }
#endif
    // The workgroup's last wavefront adds the workgroup's count of settled pixels to the statistics (spread over 256 words:
    // a single hot word serialises in L2) and takes the workgroup's end-of-call ticket.
    const uint32_t fx_settled = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(fx_lane), 63);
    uint32_t last_of_wg = 0;
    if (tid == 0) {
        if (fx_settled) atomicAdd(&s_wg[1], fx_settled);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        last_of_wg = atomicAdd(&s_wg[0], 1u) == (uint32_t)K4_WAVES - 1 ? 1u : 0u;
    }
    if (!__builtin_amdgcn_readfirstlane((int)last_of_wg)) return;
    uint32_t dep = 0;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t n = s_wg[1];
        if (n && p.stats) dep = atomicAdd(&p.stats[blockIdx.x & 255], n);
    }
    status_epilogue(p.status, p.h_status, gridDim.x, p.keep_status, dep);
}

// ---- 4:2:0 (extension): fast path ------------------------------------------------------------------------------------------
// What k_idct_colour_fast does for 4:4:4, for 16x16 MCUs of six blocks (Y00 Y01 Y10 Y11 Cb Cr, dense coefficient layout, a bound per
// block from K2 as there).  One wavefront per tile of four MCUs (64 x 16 pixels), three passes of eight blocks, eight lanes per block as
// in k_idct_colour_fast (block_fast: lane j of a group ends with sample row j of its block): first the tile's eight chroma blocks --
// their rounded samples, and per sample whether it is within the block's bound of a rounding boundary, go to LDS -- then the luma
// blocks of MCUs 0, 1 and those of MCUs 2, 3: lane j of luma block (by, bx) holds pixel row by * 8 + j, columns bx * 8 .. bx * 8 + 7
// of its MCU, and reads chroma row (by * 8 + j) / 2, columns bx * 4 .. bx * 4 + 3 (a chroma sample covers 2 x 2 pixels, no
// interpolation: k_idct_colour_exact_420 above is the definition).  The f32 colour arithmetic and its keys are k_idct_colour_fast's.
// A pixel that has an untrusted sample (luma or chroma) or G term, and every pixel of an MCU whose chroma may leave the f32 colour
// arithmetic's range, is settled at once in the finished tile in LDS, before the tile is written back: the lane evaluates the samples
// concerned in the reference's order from the dense coefficients (exact_sample_lane) and converts with colour_exact.  No queue across
// tiles: a marked pass costs a divergent round or two, which this extension can afford (the reference-order kernel it replaces
// evaluates every sample that way).
struct Idct420Params {
    const int16_t* coef;    // [mcu][6][64] natural order
    const float* ebound;    // [mcu][6] (block_ebound: magnitude = bound, lowest mantissa bit of a chroma block's = samples may leave +-249)
    uint8_t* rgb;           // the picture padded to whole MCUs
    uint32_t mcus_w, mcus_h, pitch, tiles_w, ntiles;
    uint32_t* stats;        // [256] counters of pixels settled in the reference's order (may be null)
};
constexpr int T420_STRIDE = 208;   // bytes per tile row in LDS (192 + padding, as TILE_ROW_STRIDE)

__global__ __launch_bounds__(256) void k_idct_colour_fast_420(Idct420Params p, QTables qt)
{
    __shared__ __attribute__((aligned(16))) float s_m[2][64];          // AC input scales, natural order
    __shared__ __attribute__((aligned(16))) uint32_t s_qi[2 * 64];     // quantisers, natural order
    __shared__ double s_cos[64];
    __shared__ __attribute__((aligned(16))) float s_chr_all[4][8][8][8];   // [wavefront][chroma block of the tile][row][column] rounded samples (minus the level shift)
    __shared__ uint32_t s_chf_all[4][64];                                  // [..][block * 8 + row] bit i: column i is untrusted; bit 8: the block's samples may leave the colour range
    __shared__ __attribute__((aligned(16))) uint8_t s_tile_all[4][16 * T420_STRIDE];

    const int tid = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane8 = tid & 7, grp = tid >> 3;
    const int u = lane8 < 4 ? 2 * lane8 : 2 * (lane8 - 4) + 1;   // coefficient row this lane loads
    if (threadIdx.x < 128) {
        const int t = threadIdx.x >> 6, k = threadIdx.x & 63;
        s_m[t][k] = 0.25f * cc_of(k >> 3, k & 7) * (float)qt.q[t][k];
        s_qi[t * 64 + k] = qt.q[t][k];
    } else if (threadIdx.x < 192) {
        s_cos[tid] = c_cos[tid];
    }
    LaneConst lc;
    lc.q0[0] = (float)qt.q[0][u * 8];
    lc.q0[1] = (float)qt.q[1][u * 8];
    lc.cc0 = cc_of(u, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // the lane's column-pass constants (see k_idct_colour_fast)
        float r = cosf_tab(1 * (2 * k));
#pragma unroll
        for (int l = 1; l < 8; ++l) {
            const float c = l < 4 ? cosf_tab((2 * l + 1) * (2 * k)) : -cosf_tab((2 * (7 - l) + 1) * (2 * k + 1));
            r = lane8 == l ? c : r;
        }
        lc.k[k] = r;
    }
    lc.s = lane8 < 4 ? -1.0f : 1.0f;
    __syncthreads();
    const uint32_t tile = blockIdx.x * 4u + (uint32_t)wave;
    if (tile >= p.ntiles) return;   // wave-uniform (no barrier below)
    float (*s_chr)[8][8] = s_chr_all[wave];
    uint32_t* const s_chf = s_chf_all[wave];
    uint8_t* const s_tile = s_tile_all[wave];
    const uint32_t trow = tile / p.tiles_w, tcol = tile - trow * p.tiles_w;
    const uint32_t m0 = tcol * 4u, nm = min(4u, p.mcus_w - m0);
    const size_t mcu0 = (size_t)trow * p.mcus_w + m0;

    // ---- the tile's chroma blocks: group g = block (g & 1 ? Cr : Cb) of MCU g >> 1
    {
        const uint32_t mt = (uint32_t)grp >> 1, mte = mt < nm ? mt : 0u;   // (MCUs beyond the picture read the tile's first one: never stored)
        const size_t blk = (mcu0 + mte) * 6 + 4 + ((uint32_t)grp & 1u);
        const uint4 d = reinterpret_cast<const uint4*>(p.coef + blk * 64)[u];
        const float e = p.ebound[blk];
        const bool big = __ballot(((d.z | d.w) | (u >= 4 ? (d.x | d.y) : 0u)) != 0) != 0;
        float v[8];
        if (big) block_fast<8>(d, lc, &s_m[1][u * 8], 1, v);
        else block_fast<4>(d, lc, &s_m[1][u * 8], 1, v);
        const float thr = 0.5f - fabsf(e);   // |v - rint v| >= thr: within the block's bound of a rounding boundary (a NaN counts)
        uint32_t un = 0;
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            r[i] = __builtin_rintf(v[i]);
            un |= !(fabsf(v[i] - r[i]) < thr) ? 1u << i : 0u;
        }
        float4* dst = reinterpret_cast<float4*>(&s_chr[grp][lane8][0]);
        dst[0] = make_float4(r[0], r[1], r[2], r[3]);
        dst[1] = make_float4(r[4], r[5], r[6], r[7]);
        s_chf[grp * 8 + lane8] = un | ((__float_as_uint(e) & 1u) << 8);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    uint32_t settled = 0;
    // ---- the luma blocks, two MCUs a pass: group g = block g & 3 (Y00 Y01 Y10 Y11) of MCU 2 * pass + (g >> 2)
#pragma unroll 1
    for (uint32_t pass = 0; pass < 2; ++pass) {
        const uint32_t mt = pass * 2u + ((uint32_t)grp >> 2), mte = mt < nm ? mt : 0u;
        const uint32_t yb = (uint32_t)grp & 3u, by = yb >> 1, bx = yb & 1u;
        const size_t mcu = mcu0 + mte;
        const uint4* const yrows = reinterpret_cast<const uint4*>(p.coef + (mcu * 6 + yb) * 64);
        const uint4 d = yrows[u];
        const float e = p.ebound[mcu * 6 + yb];
        const bool big = __ballot((d.w | (u >= 6 ? (d.x | d.y | d.z) : 0u)) != 0) != 0;
        float vy[8];
        if (big) block_fast<8>(d, lc, &s_m[0][u * 8], 0, vy);
        else block_fast<6>(d, lc, &s_m[0][u * 8], 0, vy);
        const float thr = 0.5f - fabsf(e);
        // the chroma samples of this lane's pixel row
        const uint32_t cyr = by * 4u + ((uint32_t)lane8 >> 1);
        const float4 cb4 = *reinterpret_cast<const float4*>(&s_chr[mte * 2][cyr][bx * 4]);
        const float4 cr4 = *reinterpret_cast<const float4*>(&s_chr[mte * 2 + 1][cyr][bx * 4]);
        const uint32_t fcb = s_chf[(mte * 2) * 8 + cyr], fcr = s_chf[(mte * 2 + 1) * 8 + cyr];
        const float cbv[4] = {cb4.x, cb4.y, cb4.z, cb4.w}, crv[4] = {cr4.x, cr4.y, cr4.z, cr4.w};
        uint32_t pk[6] = {0, 0, 0, 0, 0, 0};
        uint32_t my = 0, mg = 0;   // bit i: pixel column i has an untrusted luma sample / G term
        float yo_prev = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float ry = __builtin_rintf(vy[i]);
            my |= !(fabsf(vy[i] - ry) < thr) ? 1u << i : 0u;
            const float rb = cbv[i >> 1], rr = crv[i >> 1];
            // the colour arithmetic of k_idct_colour_fast's pixel loop (ranges and rounding argument: there, tests/test_tables.py)
            const float yo = ry + 127.501f;
            const float R = __builtin_fmaf(rr, 1.402f, yo);
            const float B = __builtin_fmaf(rb, 1.772f, yo);
            const float t = __builtin_fmaf(rr, 0.714136f, rb * 0.344136f);
            const float tc = ceilf(t);
            const float G = yo - tc;
            if (!(i & 1)) {   // (t belongs to the pixel pair)
                const bool gun = !(fabsf((tc - t) - 0.5f) < (0.5f - KPEG_G_DELTA)) && !(fabsf(t) < 4.0e-6f);   // within DELTA of a non-zero integer
                mg |= gun ? 3u << i : 0u;
            }
            (void)yo_prev;
            pk[(3 * i) >> 2] = pk_u8(R, (3 * i) & 3, pk[(3 * i) >> 2]);
            pk[(3 * i + 1) >> 2] = pk_u8(G, (3 * i + 1) & 3, pk[(3 * i + 1) >> 2]);
            pk[(3 * i + 2) >> 2] = pk_u8(B, (3 * i + 2) & 3, pk[(3 * i + 2) >> 2]);
        }
        uint8_t* const trow_lds = s_tile + (by * 8u + (uint32_t)lane8) * T420_STRIDE + (mt * 16u + bx * 8u) * 3u;
        {
            uint2* dst = reinterpret_cast<uint2*>(trow_lds);
            dst[0] = make_uint2(pk[0], pk[1]);
            dst[1] = make_uint2(pk[2], pk[3]);
            dst[2] = make_uint2(pk[4], pk[5]);
        }
        // a chroma sample's flag covers two pixel columns
        auto spread = [](uint32_t n4) -> uint32_t { return ((n4 & 1u) * 3u) | ((n4 & 2u) * 6u) | ((n4 & 4u) * 12u) | ((n4 & 8u) * 24u); };
        const uint32_t mcb = spread((fcb >> (bx * 4u)) & 15u), mcr = spread((fcr >> (bx * 4u)) & 15u);
        const bool wide = (((fcb | fcr) >> 8) & 1u) != 0;
        uint32_t todo = mt < nm ? (wide ? 0xFFu : (my | mg | mcb | mcr)) : 0u;
        if (__ballot(todo != 0)) {   // wave-uniform
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            settled += (uint32_t)__popc(todo);
            const uint4* const cbrows = reinterpret_cast<const uint4*>(p.coef + (mcu * 6 + 4) * 64);
            const uint4* const crrows = reinterpret_cast<const uint4*>(p.coef + (mcu * 6 + 5) * 64);
            while (__ballot(todo != 0)) {   // wave-uniform: a marked pixel of every lane per round
                if (todo) {
                    const uint32_t i = 31u - (uint32_t)__builtin_clz(todo);
                    todo &= ~(1u << i);
                    float fy = vy[0], fb = cbv[0], fr = crv[0];
#pragma unroll
                    for (int k = 1; k < 8; ++k) fy = i == (uint32_t)k ? vy[k] : fy;
#pragma unroll
                    for (int k = 1; k < 4; ++k) fb = (i >> 1) == (uint32_t)k ? cbv[k] : fb, fr = (i >> 1) == (uint32_t)k ? crv[k] : fr;
                    int Sy = (int)__builtin_rintf(fy) + 128, Sb = (int)fb + 128, Sr = (int)fr + 128;
                    const int cx = (int)cyr, cy = (int)(bx * 4u + (i >> 1));
                    if ((my >> i) & 1u) Sy = exact_sample_lane(yrows, s_qi, s_cos, lane8, (int)i);
                    if ((mcb >> i) & 1u) Sb = exact_sample_lane(cbrows, s_qi + 64, s_cos, cx, cy);
                    if ((mcr >> i) & 1u) Sr = exact_sample_lane(crrows, s_qi + 64, s_cos, cx, cy);
                    const uint32_t px = colour_exact(Sy, Sb, Sr);
                    trow_lds[i * 3] = (uint8_t)px, trow_lds[i * 3 + 1] = (uint8_t)(px >> 8), trow_lds[i * 3 + 2] = (uint8_t)(px >> 16);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- write the tile back: 16 rows x nm * 48 bytes
    uint8_t* const base = p.rgb + (size_t)trow * 16 * p.pitch + (size_t)m0 * 48;
    if (nm == 4 && ((reinterpret_cast<uintptr_t>(p.rgb) | p.pitch) & 15) == 0) {
#pragma unroll
        for (uint32_t c = (uint32_t)tid; c < 16 * 12; c += 64) {
            const uint32_t r = c / 12u, k = c - r * 12u;
            __builtin_nontemporal_store(*reinterpret_cast<const uint4v*>(s_tile + r * T420_STRIDE + k * 16), reinterpret_cast<uint4v*>(base + (size_t)r * p.pitch + k * 16));
        }
    } else {
        const uint32_t per_row = nm * 12;   // 4-byte pieces
        for (uint32_t c = (uint32_t)tid; c < 16 * per_row; c += 64) {
            const uint32_t r = c / per_row, k = c - r * per_row;
            *reinterpret_cast<uint32_t*>(base + (size_t)r * p.pitch + k * 4) = *reinterpret_cast<const uint32_t*>(s_tile + r * T420_STRIDE + k * 4);
        }
    }
    if (p.stats) {
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(settled), 63);
        if (n && tid == 0) atomicAdd(&p.stats[blockIdx.x & 255], n);
    }
}


}  // namespace kpeg_dev
