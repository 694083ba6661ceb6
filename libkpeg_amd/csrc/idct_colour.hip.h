// libkpeg_amd/csrc/idct_colour.hip.h -- K4: dequantise + 8x8 IDCT + level shift + YCbCr->RGB
// + MCU tiling, for gfx950 (MI355X).
//
// Replaces MCU::constructMCU's dequantisation (src/MCU.cpp:110-120), MCU::computeIDCT
// (:172-216), performLevelShift (:218-245), convertYCbCrToRGB (:247-279) and
// Image::createImageFromMCUs (src/Image.cpp:20-86) of the reference.
//
// Bit-exactness.  The reference evaluates every sample as a 64-term sum accumulated in
// *float* in (u outer, v inner) order from double products; its rounding
// cannot be reproduced by a fast transform.  The kernel therefore computes
//   (1) a fast separable f32 IDCT whose distance to the reference's float result is
//       bounded rigorously per block (tools/idct_bound.py derives the constant), and
//   (2) only where the fast value lies within that bound of a rounding boundary (or the G
//       term is too close to an integer for the f32 colour arithmetic), the reference-order
//       evaluation itself: such pixels are queued in the tile loop and fixed up in passes,
//       after their tile has been stored (see k_idct_colour_fast).
// Blocks with no AC coefficient are exact in (1) by construction.
//
// Mapping (no MFMA: byte/short work, not a dense contraction; the kernel is bound by VALU issue,
// and on gfx950 only add/sub/mul/fma/and/or/mov issue at 1.7 cycles, everything else at 2.7:
// tools/ubench/valu_rate.hip).
//   * one wavefront per workgroup, 8 lanes per MCU, 8 MCUs (64x8 pixels) per tile; the grid is the
//     resident wavefronts, each walks tiles blockIdx.x, + gridDim.x, ...
//   * lane j of an 8-lane group loads one 16-byte row of each component block
//     (rows 0,2,4,6 on lanes 0-3, rows 1,3,5,7 on lanes 4-7): a wavefront's three
//     global_load_dwordx4 cover 8 MCUs x 384 B = 3 KiB of contiguous coefficients.
//   * row pass (over v) in registers: even/odd decomposition, 34 f32 ops per 8 samples;
//     column pass (over u) across the 8 lanes with DPP: quad broadcasts feed 4-term
//     even (lanes 0-3) / odd (lanes 4-7) sums, one row_half_mirror FMA combines them.
//     Lane l ends up with pixel row l of the block: 8 pixels x 3 components.
//   * colour conversion in f32/int-exact arithmetic (proven ranges), RGB bytes staged in an
//     LDS tile (8 rows x 768 B, padded rows) and written back with 16-byte coalesced stores.
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
//   A >= KPEG_A_LIM (any component): beyond the range the colour arithmetic's rounding argument is checked for
//       (|sample| <= 4100, tests/test_tables.py): E = +inf, all the block's samples take the reference-order path.
//   A >= KPEG_A_LIM_CHROMA (chroma block): the samples may leave the range the f32 colour arithmetic is proven
//       for (|.| < 250): the lowest mantissa bit of E is set (E is first rounded up to an even mantissa, so the
//       bit never shrinks it) and K4 converts that MCU's pixels with the reference's double arithmetic in-lane.
//       Saturated colour edges do this in photographs; dense noise does it everywhere.
#define KPEG_A_LIM_CHROMA 249.0f
#define KPEG_A_LIM 4000.0f
// The sign bit carries one more fact about the block: set = every non-zero AC coefficient sits at
// (0,1), (1,0) or (1,1).  Those blocks produce nearly all true ties (equal and opposite (0,1)/(1,0)
// terms cancel on the diagonal), and their reference-order sum has at most four terms, which the
// lane that holds the pixel's queue entry evaluates itself in the fix-up pass (exact_corner).
__device__ __forceinline__ float block_ebound(float A, int nnz_ac, bool chroma, bool corner_only)
{
    if (!(A < KPEG_A_LIM)) return __builtin_inff();
    const float E = nnz_ac ? (KPEG_U * A) * ((float)nnz_ac + KPEG_KAPPA) : 0.0f;
    uint32_t bits = (__float_as_uint(E) + 1u) & ~1u;
    if (chroma && !(A < KPEG_A_LIM_CHROMA)) bits |= 1u;
    const float Ef = __uint_as_float(bits);
    return corner_only ? -Ef : Ef;
}

// E for caller-supplied coefficients (kpeg_hip_idct_colour): one thread per block.
__global__ __launch_bounds__(256) void k_ebound(const int16_t* coef, uint32_t nblocks, QTables qt, float* ebound)
{
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= nblocks) return;
    const int t = (b % 3) ? 1 : 0;
    const uint4* src = reinterpret_cast<const uint4*>(coef + (size_t)b * 64);
    float A = 0.f;
    int n = 0;
    bool corner = true;
    for (int r = 0; r < 8; ++r) {
        const uint4 d = src[r];
        const uint32_t w[4] = {d.x, d.y, d.z, d.w};
        for (int i = 0; i < 8; ++i) {
            const int c = (int)(short)((w[i >> 1] >> ((i & 1) * 16)) & 0xFFFF);
            const int k = r * 8 + i;
            if (k == 0) {
                A += fabsf(0.25f * (cc_of(0, 0) * ((float)c * (float)qt.q[t][0])));
            } else if (c != 0) {
                A += fabsf((float)c * (0.25f * cc_of(r, i) * (float)qt.q[t][k]));
                n++;
                corner = corner && (k == 1 || k == 8 || k == 9);
            }
        }
    }
    ebound[b] = block_ebound(A, n, t != 0, corner);
}

typedef unsigned int uint3v __attribute__((ext_vector_type(3)));
typedef unsigned int __attribute__((ext_vector_type(4), may_alias)) uint4v;  // 16-byte view of uint32_t LDS words
constexpr int TILE_MCUS = 8;                    // MCUs per wavefront iteration (8 lane groups)
constexpr int TILE_ROW_BYTES = TILE_MCUS * 24;  // 192
constexpr int TILE_ROW_STRIDE = 208;            // padded: 13 x 16 bytes, conflict-free b64 writes across rows
constexpr int QUEUE_CAP = 64;                   // queue entries per wavefront (fix-ups run from QUEUE_FLUSH entries on); pixels beyond go to the overflow list
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

// ---- reference-order evaluation of single samples (the fix-up passes) ---------------------------
// Value of one sample of a block whose non-zero coefficients all lie in the 2x2 low-frequency corner:
// MCU::computeIDCT's sum (MCU.cpp:184-198) restricted to the terms (0,0), (0,1), (1,0), (1,1) in that order --
// the others are zero and leave the float accumulator unchanged, as do zero terms among these four
// (x + (+-0) == x), so no test is needed.  cos((2x+1)*0*pi/16) == 1.0 exactly.
//   w0 / w1: the words holding coefficients (0,0),(0,1) / (1,0),(1,1); q..: the four quantisers;
//   cx1 = cosT[x][1], cy1 = cosT[y][1].  Returns roundl(ic) as a float (the sample minus the level shift).
// One lane per sample: nearly all unsafe pixels are of this kind (structural ties, see the kernel).
__device__ __forceinline__ float exact_corner(uint32_t w0, uint32_t w1, uint32_t q00, uint32_t q01, uint32_t q10, uint32_t q11,
                                              double cx1, double cy1)
{
    const float c0 = 0x1.6a09e6p-1f;  // (float)(1/sqrt 2)
    const int F00 = (int)(short)(w0 & 0xFFFF) * (int)q00, F01 = ((int)w0 >> 16) * (int)q01;
    const int F10 = (int)(short)(w1 & 0xFFFF) * (int)q10, F11 = ((int)w1 >> 16) * (int)q11;
    const float fc00 = (c0 * c0) * (float)F00, fc01 = (c0 * 1.0f) * (float)F01, fc10 = (1.0f * c0) * (float)F10,
                fc11 = (float)F11;
    float sum = fc00;                                              // (float)(0.0 + fc00 * 1.0 * 1.0)
    sum = (float)((double)sum + (double)fc01 * cy1);              // ((double)fc01 * 1.0) * cy1
    sum = (float)((double)sum + (double)fc10 * cx1);              // ((double)fc10 * cx1) * 1.0
    sum = (float)((double)sum + ((double)fc11 * cx1) * cy1);
    const float ic = (float)(0.25 * (double)sum);
    const float t = truncf(ic), fr = ic - t;
    return t + (fr >= 0.5f ? 1.0f : 0.0f) - (fr <= -0.5f ? 1.0f : 0.0f);  // roundl: half away from zero
}

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

#ifndef KPEG_PUSH_GROUP
#define KPEG_PUSH_GROUP 4           // pixel columns per unsafe-pixel test, dense layout (measured 1 / 2 / 4: 68.9 / 65.3 / 64.0 us)
#endif
#ifndef KPEG_PUSH_GROUP_COMPACT
#define KPEG_PUSH_GROUP_COMPACT 2   // ... compact stream (69.0 / 68.1 / 71.3 us: four keep too many registers live there)
#endif
#ifndef KPEG_K4_STASH_DENSE
#define KPEG_K4_STASH_DENSE 0   // dense layout: 1 = queue entries carry their blocks' corner words too (six lane shuffles per tile and six
                                // more VGPRs: measured no gain); 0 = the fix-up pass loads them from the coefficient buffer
#endif
#ifndef KPEG_K4_CHROMA2
#define KPEG_K4_CHROMA2 0
#endif
#ifndef KPEG_K4_LUMA4
#define KPEG_K4_LUMA4 0
#endif
#ifndef KPEG_QUEUE_FLUSH
#define KPEG_QUEUE_FLUSH 32
#endif
constexpr int QUEUE_FLUSH = KPEG_QUEUE_FLUSH;   // queued pixels that make a fix-up pass worth its fixed cost
constexpr int OVER_CAP = TILE_MCUS * 64 - (QUEUE_CAP - QUEUE_FLUSH);   // a tile starts with at least QUEUE_CAP - QUEUE_FLUSH free entries
constexpr int QUEUE_WORDS = 8;    // per queued pixel: position, 3 rounded samples, 3 keys (>= 0: that component is unsafe), pad
constexpr int QUEUE_WORDS_COMPACT = 16;   // ... + words 8..13: the 2x2 corner coefficients of the pixel's three blocks (the compact
                                          // stream has no block to read them from later)
constexpr int IMG_BYTES = 24 * 128;       // compact path: the tile's 24 blocks rebuilt in LDS, natural order, int16

#ifdef KPEG_K4_STAMP
__device__ unsigned long long g_k4_stamp[8192 * 4];
#endif
// One workgroup of K4_WAVES wavefronts per CU; every wavefront is an independent worker on tiles of 8 MCUs (64 x 8
// pixels), no barrier after the start-up.  The workgroup owns a contiguous range of tiles and hands them out through a
// counter in LDS: on a SIMD the oldest wavefront gets the issue slots first (age arbitration), so with a static split the
// wavefronts of one SIMD finished one after the other -- first 38 us, last 54-66 us, the SIMD two-thirds idle at the end
// (profiles/r02: per-wavefront stamps) -- whereas wavefronts that take tiles as they go all finish together.
//
// Pixels whose fast value cannot be trusted (within the block's bound of a rounding boundary, or a G term too
// close to an integer) are only *noted* in the tile loop: their position goes to a small queue in LDS.  Once the
// tiles they belong to have been written out, a fix-up pass evaluates them in the reference's own order, one lane
// per (pixel, component), and patches the three bytes in global memory.  Handling them where they are found -- a few
// lanes of a wavefront, several times per tile -- cost 37 % of the kernel's time (profiles/r01_g: 0.103 -> 0.065 ms with
// the handling compiled out).
template <bool COMPACT>
__global__ __launch_bounds__(K4_THREADS) void k_idct_colour_fast(IdctParams p, QTables qt)
{
    constexpr int QW = (COMPACT || KPEG_K4_STASH_DENSE) ? QUEUE_WORDS_COMPACT : QUEUE_WORDS;   // entries that stash the blocks' corner words are twice as long
    __shared__ __attribute__((aligned(16))) uint8_t s_tile_all[K4_WAVES][8 * TILE_ROW_STRIDE];
    __shared__ __attribute__((aligned(16))) uint32_t s_queue_all[K4_WAVES][QUEUE_CAP * QW];
    __shared__ __attribute__((aligned(16))) uint32_t s_img_all[COMPACT ? K4_WAVES : 1][IMG_BYTES / 4];
    __shared__ uint16_t s_over_all[K4_WAVES][OVER_CAP];   // unsafe pixels a tile has beyond the queue's room: bits [11:0] of the position word (the
                                            // fix-up pass that takes them runs before the next tile: the tile is known)
    __shared__ __attribute__((aligned(16))) float s_m[2][64];     // AC input scales, natural order
    __shared__ __attribute__((aligned(16))) uint32_t s_qi[2][64]; // quantisers (exact dequantisation in the fix-up pass)
    __shared__ double s_cos[64];
    __shared__ uint32_t s_next;       // next tile of this workgroup's range to hand out
    __shared__ uint32_t s_wg[2];      // [0] wavefronts of this workgroup that are done, [1] unsafe pixels they counted

#ifdef KPEG_K4_STAMP
    // diagnostic build only (tools/k4_clock.py): the shader clock this kernel runs at = d(s_memtime) / d(s_memrealtime) x 100 MHz
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x & 63;   // lane of the wavefront
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint8_t* const s_tile = s_tile_all[wave];
    uint32_t* const s_queue = s_queue_all[wave];
    uint32_t* const s_img = s_img_all[COMPACT ? wave : 0];
    uint16_t* const s_over = s_over_all[wave];
    const int lane8 = tid & 7;          // lane within the MCU group = output pixel row
    const int grp = tid >> 3;           // MCU within the tile, 0..7
    const int u = lane8 < 4 ? 2 * lane8 : 2 * (lane8 - 4) + 1;  // coefficient row this lane loads

    if (threadIdx.x < 128) {
        const int t = threadIdx.x >> 6, k = threadIdx.x & 63;
        s_m[t][k] = 0.25f * cc_of(k >> 3, k & 7) * (float)qt.q[t][k];
        s_qi[t][k] = qt.q[t][k];
    } else if (threadIdx.x < 192) {
        s_cos[tid] = c_cos[tid];
    } else if (threadIdx.x == 192) {
        s_next = 0;
        s_wg[0] = 0;
        s_wg[1] = 0;
    }
    // this workgroup's tiles: [wg_tile0, wg_tile0 + wg_ntiles)
    const uint32_t wg_tile0 = (uint32_t)(((unsigned long long)p.ntiles * blockIdx.x) / gridDim.x);
    const uint32_t wg_ntiles = (uint32_t)(((unsigned long long)p.ntiles * (blockIdx.x + 1)) / gridDim.x) - wg_tile0;
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
    const uint32_t wb_r = (uint32_t)tid / 12u, wb_k = (uint32_t)tid - wb_r * 12u;
    const uint32_t wb_lds = wb_r * TILE_ROW_STRIDE + wb_k * 16, wb_g = wb_r * p.pitch + wb_k * 16;
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
        uint32_t oA = wb_g;
        asm volatile("" : "+v"(oA));
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

    uint32_t dbg_count = 0;   // experiments (KPEG_COUNT_*): reported in place of the unsafe-pixel count

    // Fix-up passes.  A queue entry is a pixel of one of this wavefront's tiles:
    //   word 0: [2:0] MCU within the tile, [5:3] pixel row, [8:6] pixel column, [11:9] component blocks that are
    //           corner-only (the sign of their bound), [31:12] the tile's number k inside the workgroup's range (tile = wg_tile0 + k);
    //   words 1..3: the three rounded fast samples (minus the level shift); words 4..6: their keys (>= 0: unsafe).
    // nq queued entries, then nover pixels of the overflow list: position words only (a tile with more unsafe pixels
    // than the queue had room for: clusters of ties, adversarial input), all components to be evaluated.  One lane per entry.  Unsafe components of corner-only blocks are settled by the lane itself
    // with the four-term sum; the others (a few per wavefront) by the whole wavefront, one after the other; the lane
    // then converts and stores its pixel.
    // The tiles concerned have been written out by this wavefront before.
    auto run_fixups = [&](uint32_t nq, uint32_t nover, uint32_t over_k) {
        uint32_t lane = (uint32_t)tid;
        asm volatile("" : "+v"(lane));   // nothing of a fix-up pass is to be computed ahead of the tile loop and kept in registers
        const uint32_t total = nq + nover;
        const float ccl = cc_of((int)(lane >> 3), (int)(lane & 7));
        for (uint32_t base = 0; base < total; base += 64) {
            const uint32_t e = base + lane;
            const bool valid = e < total;
            uint32_t pos = 0;
            float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, f0 = -1.0f, f1 = -1.0f, f2 = -1.0f;
            if (valid) {
                if (e < nq) {
                    const uint4v a = *reinterpret_cast<const uint4v*>(s_queue + e * QW);
                    const uint4v b = *reinterpret_cast<const uint4v*>(s_queue + e * QW + 4);
                    pos = a.x;
                    r0 = __uint_as_float(a.y), r1 = __uint_as_float(a.z), r2 = __uint_as_float(a.w);
                    f0 = __uint_as_float(b.x), f1 = __uint_as_float(b.y), f2 = __uint_as_float(b.z);
                } else {
                    pos = (uint32_t)s_over[e - nq] | (over_k << 12);
                    f0 = f1 = f2 = 1.0f;
                }
            }
            const uint32_t g = pos & 7u, x = (pos >> 3) & 7u, y = (pos >> 6) & 7u;
            const uint32_t tile = wg_tile0 + (pos >> 12);
            const uint32_t trow = tile_row(p, tile), tcol = tile - trow * p.tiles_w;
            const uint32_t m0 = tcol * TILE_MCUS;
            const uint32_t mcu = trow * p.mcus_w + m0 + g;
            const bool need0 = f0 >= 0.0f, need1 = f1 >= 0.0f, need2 = f2 >= 0.0f;
            // corner-only blocks: the two words such a block consists of
#ifdef KPEG_FX_SKIP_CORNER
            const bool cn0 = false, cn1 = false, cn2 = false;
#else
            // A queue entry carries the corner words of its three blocks (stashed when it was pushed: no load, no memory latency
            // in this pass); an overflow-list pixel has none: dense layout -> two 4-byte loads per block, compact stream -> the
            // general way (from the tile's image).
            const bool from_queue = valid && e < nq && (COMPACT || KPEG_K4_STASH_DENSE);
            const bool stashed = !COMPACT || from_queue;
            const bool cn0 = need0 && (pos & (1u << 9)) && stashed, cn1 = need1 && (pos & (1u << 10)) && stashed, cn2 = need2 && (pos & (1u << 11)) && stashed;
#endif
            uint32_t w00 = 0, w01 = 0, w10 = 0, w11 = 0, w20 = 0, w21 = 0;
            if (from_queue) {
                const uint4v cw = *reinterpret_cast<const uint4v*>(s_queue + e * QW + 8);
                const uint2 cw2 = *reinterpret_cast<const uint2*>(s_queue + e * QW + 12);
                w00 = cw.x, w01 = cw.y, w10 = cw.z, w11 = cw.w, w20 = cw2.x, w21 = cw2.y;
            } else if constexpr (!COMPACT) {
                const uint32_t* c32 = reinterpret_cast<const uint32_t*>(p.coef) + (size_t)mcu * 96;
                if (cn0) w00 = c32[0], w01 = c32[4];
                if (cn1) w10 = c32[32], w11 = c32[36];
                if (cn2) w20 = c32[64], w21 = c32[68];
            }
            // the others: wave-uniform lists of (lane, component)
            unsigned long long g0 = __ballot(need0 && !cn0), g1 = __ballot(need1 && !cn1), g2 = __ballot(need2 && !cn2);
#ifdef KPEG_FX_SKIP_COOP
            g0 = g1 = g2 = 0;
#endif
            auto pop = [](unsigned long long& m0_, unsigned long long& m1_, unsigned long long& m2_, uint32_t& c, uint32_t& L) {
                unsigned long long& mm = m0_ ? m0_ : (m1_ ? m1_ : m2_);
                c = m0_ ? 0u : (m1_ ? 1u : 2u);
                L = (uint32_t)__builtin_ctzll(mm);
                mm &= mm - 1;
            };
            // The coefficient at position `lane` of block c of MCU mcuL (wave-uniform arguments), for the samples the whole
            // wavefront evaluates.  Dense layout: one coalesced 128-byte load.  Compact stream: from the tile's image in LDS --
            // only pixels of the tile that was computed last get here (the overflow list, settled before the next tile;
            // queued pixels had their non-corner samples settled by resolve_noncorner while their tile's image stood).
            auto fetch_coef = [&](uint32_t mcuL, uint32_t c, int) -> int {
                if constexpr (!COMPACT) {
                    return p.coef[((size_t)mcuL * 3 + c) * 64 + lane];
                } else {
                    const uint16_t* img16 = reinterpret_cast<const uint16_t*>(s_img);
                    return (int)(int16_t)img16[((mcuL & (TILE_MCUS - 1)) * 3 + c) * 64 + lane];
                }
            };
#ifndef KPEG_COOP_BATCH
#define KPEG_COOP_BATCH 2
#endif
            constexpr int BATCH = KPEG_COOP_BATCH;
            const bool many = __popcll(g0) + __popcll(g1) + __popcll(g2) > BATCH;   // wave-uniform
#if defined(KPEG_COUNT_COOP)
            dbg_count += __popcll(g0) + __popcll(g1) + __popcll(g2);
#elif defined(KPEG_COUNT_MANY)
            dbg_count += many ? 1u : 0u;
#elif defined(KPEG_COUNT_FLUSH)
            dbg_count += 1u;
#elif defined(KPEG_COUNT_CORNER)
            dbg_count += __popcll(__ballot(cn0)) + __popcll(__ballot(cn1)) + __popcll(__ballot(cn2));
#endif
            // a batch of those: one coalesced 128-byte load per sample, all in flight together with the corner words
            // (one memory latency per pass -- at the end of a wavefront's life nothing hides it)
            int cf[BATCH];
            unsigned long long a0 = g0, a1 = g1, a2 = g2;
            if (!many) {
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    cf[k] = 0;
                    if (a0 | a1 | a2) {
                        uint32_t c, L;
                        pop(a0, a1, a2, c, L);
                        const uint32_t mcuL = (uint32_t)__builtin_amdgcn_readlane((int)mcu, (int)L);
                        cf[k] = fetch_coef(mcuL, c, k);
                    }
                }
            }
            const double cx1 = s_cos[x * 8 + 1], cy1 = s_cos[y * 8 + 1];
            if (cn0) r0 = exact_corner(w00, w01, s_qi[0][0], s_qi[0][1], s_qi[0][8], s_qi[0][9], cx1, cy1);
            if (cn1) r1 = exact_corner(w10, w11, s_qi[1][0], s_qi[1][1], s_qi[1][8], s_qi[1][9], cx1, cy1);
            if (cn2) r2 = exact_corner(w20, w21, s_qi[1][0], s_qi[1][1], s_qi[1][8], s_qi[1][9], cx1, cy1);
            if (many) {
                // a tile evaluated as a whole, a cluster of unsafe pixels: every lane its own samples, component by component
#pragma unroll 1
                for (int c = 0; c < 3; ++c) {
                    const bool nd = c == 0 ? (need0 && !cn0) : (c == 1 ? (need1 && !cn1) : (need2 && !cn2));
                    if (__ballot(nd) == 0) continue;
                    int S = 128;
                    if (nd) {
                        const uint4* blk = COMPACT ? reinterpret_cast<const uint4*>(s_img) + ((mcu & (TILE_MCUS - 1)) * 3 + c) * 8
                                                   : reinterpret_cast<const uint4*>(p.coef) + ((size_t)mcu * 3 + c) * 8;
                        S = exact_sample_lane(blk, s_qi[c ? 1 : 0], s_cos, (int)x, (int)y);
                    }
                    const float rv = (float)(S - 128);
                    if (nd) {
                        if (c == 0) r0 = rv;
                        else if (c == 1) r1 = rv;
                        else r2 = rv;
                    }
                }
            } else {
                // by the whole wavefront, one sample after the other, lane = coefficient position
                for (;;) {
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) {
                        if (g0 | g1 | g2) {   // wave-uniform
                            uint32_t c, L;
                            pop(g0, g1, g2, c, L);
                            const int xL = __builtin_amdgcn_readlane((int)x, (int)L), yL = __builtin_amdgcn_readlane((int)y, (int)L);
                            const int F = cf[k] * (int)s_qi[c ? 1 : 0][lane];          // m_8x8block after MCU.cpp:110-112
                            const int S = exact_sample_wave(ccl * (float)F, s_cos, xL, yL, F != 0);
                            const float rv = (float)(S - 128);
                            if (lane == L) {
                                if (c == 0) r0 = rv;
                                else if (c == 1) r1 = rv;
                                else r2 = rv;
                            }
                        }
                    }
                    if (!(g0 | g1 | g2)) break;
                    // the next batch (clusters only)
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) {
                        cf[k] = 0;
                        if (a0 | a1 | a2) {
                            uint32_t c, L;
                            pop(a0, a1, a2, c, L);
                            const uint32_t mcuL = (uint32_t)__builtin_amdgcn_readlane((int)mcu, (int)L);
                            cf[k] = fetch_coef(mcuL, c, k);
                        }
                    }
                }
            }
            // the tile's own stores (issued before this pass's loads) must have been performed before bytes of theirs are patched
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (valid) {
                const uint32_t px = colour_exact((int)r0 + 128, (int)r1 + 128, (int)r2 + 128);
                size_t off;
                if (p.rgb_table) {
                    const uint32_t img = trow / p.rows_per_img;
                    off = (size_t)(reinterpret_cast<uintptr_t>(p.rgb_table[img]) - reinterpret_cast<uintptr_t>(p.rgb)) +
                          (size_t)(trow - img * p.rows_per_img) * 8 * p.pitch;
                } else {
                    off = (size_t)trow * 8 * p.pitch;
                }
                uint8_t* o = p.rgb + off + (size_t)x * p.pitch + (size_t)(m0 + g) * 24 + y * 3;
                o[0] = (uint8_t)px;
                o[1] = (uint8_t)(px >> 8);
                o[2] = (uint8_t)(px >> 16);
            }
        }
    };

    // Compact stream only.  The queue entries [first, last) were pushed by the tile whose image stands in LDS; those of
    // them with an unsafe sample in a block that is NOT corner-only get that sample evaluated now, in the reference's order,
    // from the image (the whole wavefront per sample, lane = coefficient position; lane-parallel when there are many) and
    // written back into the entry with its key cleared: later the compact stream offers no block to read.  What remains
    // for the deferred pass: corner-only samples (from the stashed words), the colour conversion and the three bytes.
    auto resolve_noncorner = [&](uint32_t first, uint32_t last) {
        uint32_t lane = (uint32_t)tid;
        asm volatile("" : "+v"(lane));
        const uint32_t e = first + lane;
        const bool valid = e < last;
        uint32_t pos = 0;
        float f0 = -1.0f, f1 = -1.0f, f2 = -1.0f;
        if (valid) {
            pos = s_queue[e * QW];
            const uint4v b = *reinterpret_cast<const uint4v*>(s_queue + e * QW + 4);
            f0 = __uint_as_float(b.x), f1 = __uint_as_float(b.y), f2 = __uint_as_float(b.z);
        }
        const uint32_t g = pos & 7u, x = (pos >> 3) & 7u, y = (pos >> 6) & 7u;
        const bool n0 = f0 >= 0.0f && !(pos & (1u << 9)), n1 = f1 >= 0.0f && !(pos & (1u << 10)), n2 = f2 >= 0.0f && !(pos & (1u << 11));
        unsigned long long g0 = __ballot(n0), g1 = __ballot(n1), g2 = __ballot(n2);
        if (!(g0 | g1 | g2)) return;
        float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
        const uint16_t* img16 = reinterpret_cast<const uint16_t*>(s_img);
        if (__popcll(g0) + __popcll(g1) + __popcll(g2) > 6) {
            // a cluster: every lane its own samples, component by component
#pragma unroll 1
            for (int c = 0; c < 3; ++c) {
                const bool nd = c == 0 ? n0 : (c == 1 ? n1 : n2);
                if (__ballot(nd) == 0) continue;
                int S = 128;
                if (nd) S = exact_sample_lane(reinterpret_cast<const uint4*>(s_img) + (g * 3 + c) * 8, s_qi[c ? 1 : 0], s_cos, (int)x, (int)y);
                const float rv = (float)(S - 128);
                if (c == 0) r0 = rv;
                else if (c == 1) r1 = rv;
                else r2 = rv;
            }
        } else {
            const float ccl = cc_of((int)(lane >> 3), (int)(lane & 7));
            while (g0 | g1 | g2) {   // wave-uniform
                unsigned long long& mm = g0 ? g0 : (g1 ? g1 : g2);
                const uint32_t c = g0 ? 0u : (g1 ? 1u : 2u);
                const uint32_t L = (uint32_t)__builtin_ctzll(mm);
                mm &= mm - 1;
                const uint32_t gL = (uint32_t)__builtin_amdgcn_readlane((int)g, (int)L);
                const int xL = __builtin_amdgcn_readlane((int)x, (int)L), yL = __builtin_amdgcn_readlane((int)y, (int)L);
                const int F = (int)(int16_t)img16[(gL * 3 + c) * 64 + lane] * (int)s_qi[c ? 1 : 0][lane];   // m_8x8block after MCU.cpp:110-112
                const int S = exact_sample_wave(ccl * (float)F, s_cos, xL, yL, F != 0);
                const float rv = (float)(S - 128);
                if (lane == L) {
                    if (c == 0) r0 = rv;
                    else if (c == 1) r1 = rv;
                    else r2 = rv;
                }
            }
        }
        if (n0) s_queue[e * QW + 1] = __float_as_uint(r0), s_queue[e * QW + 4] = __float_as_uint(-1.0f);   // key < 0: settled
        if (n1) s_queue[e * QW + 2] = __float_as_uint(r1), s_queue[e * QW + 5] = __float_as_uint(-1.0f);
        if (n2) s_queue[e * QW + 3] = __float_as_uint(r2), s_queue[e * QW + 6] = __float_as_uint(-1.0f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    bool have_prev = false;
    size_t prev_off = 0;
    uint32_t prev_nm = 0;
    uint32_t nq = 0;           // queued positions (wave-uniform)
    uint32_t nover = 0;        // entries of the overflow list
    uint32_t nq_total = 0;
    // A tile's inputs: one 16-byte coefficient row of each component block and the three blocks' bounds.  They are asked
    // for one tile ahead (a second register set: the 16-wavefront workgroup leaves 128 VGPRs per lane), so that a
    // wavefront's tile costs it its instructions and not a memory round trip on top -- with four wavefronts per SIMD the
    // others cannot cover that wait.
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
        const uint32_t tile = wg_tile0 + tk;
        const uint32_t a = p.tile_start[tile], b = p.tile_start[tile + 1];   // wave-uniform addresses: scalar loads
        rs = a;
        rn = (b >= a && b <= p.rec_cap && b - a <= 24u * 63u) ? b - a : 0u;   // (a corrupt stream may leave anything in the table)
    };
    auto issue_loads = [&](uint32_t tk, TileIn& in, uint32_t rs, uint32_t rn) {
        const uint32_t tile = wg_tile0 + tk;
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
            in.r0 = l < rn ? rp[l] : 31u;            // (block 31 does not exist: skipped)
            in.r1 = l + 64u < rn ? rp[l + 64u] : 31u;
            in.dcw = l < 24u ? (uint32_t)(uint16_t)p.dc16[mcu_ld * 3 + l] : 0u;
        } else {
            const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(p.coef + mcu_ld * 192) + (lane_mcu * 384u + (uint32_t)u * 16u));
#if defined(KPEG_ABLATE_HALFLINE)
            // timing experiment: rows 4..7 (the second 64 bytes of every block) are not loaded -- does half a line cost half?
            in.d0 = in.d1 = in.d2 = make_uint4(0, 0, 0, 0);
            if (u < 4) in.d0 = src[0], in.d1 = src[8], in.d2 = src[16];
#elif defined(KPEG_ABLATE_QUARTERLINE)
            in.d0 = in.d1 = in.d2 = make_uint4(0, 0, 0, 0);
            if (u < 2) in.d0 = src[0], in.d1 = src[8], in.d2 = src[16];
#else
            in.d0 = src[0], in.d1 = src[8], in.d2 = src[16];
#endif
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
        const uint32_t tilek = tilek_cur;              // this tile's number inside the workgroup's range
        const bool more = tilek < wg_ntiles;           // wave-uniform
        const uint32_t tile = wg_tile0 + tilek;
        uint32_t tilek_after = 0;
        if (more) {
        tilek_after = take_tile();
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
        }
        const float e0 = cur.e0, e1 = cur.e1, e2 = cur.e2;
        if constexpr (PREFETCH) {
            // the next tile's inputs travel while this one is computed (asked for here, behind the rebuild: this tile's
            // records are dead by now and their registers free)
            if (tilek_next < wg_ntiles) issue_loads(tilek_next, nxt, rs_next, rn_next);
            asm volatile("" ::: "memory");
        }
        // The first word of rows 0 and 1 of the MCU's three blocks (coefficients (0,0),(0,1) and (1,0),(1,1)): they sit on
        // lanes 0 and 4 of the group.  A queued pixel takes them along, so the fix-up pass settles corner-only blocks (nine
        // unsafe samples in ten) without touching memory.
        constexpr bool STASH = COMPACT || KPEG_K4_STASH_DENSE;
        const int lrow0 = tid & 56, lrow1 = lrow0 | 4;
        uint32_t cw00 = 0, cw01 = 0, cw10 = 0, cw11 = 0, cw20 = 0, cw21 = 0;
        if constexpr (STASH && !COMPACT) {   // (the compact path reads them from the tile's LDS image when it queues a pixel)
            cw00 = (uint32_t)__shfl((int)d0.x, lrow0), cw01 = (uint32_t)__shfl((int)d0.x, lrow1);
            cw10 = (uint32_t)__shfl((int)d1.x, lrow0), cw11 = (uint32_t)__shfl((int)d1.x, lrow1);
            cw20 = (uint32_t)__shfl((int)d2.x, lrow0), cw21 = (uint32_t)__shfl((int)d2.x, lrow1);
        }
        const size_t cur_off = tile_offset(trow, m0);
        if (have_prev) write_back(prev_off, prev_nm);  // LDS still holds the previous tile
        have_prev = true;
        prev_off = cur_off;
        prev_nm = nm;

        float v[3][8];
#ifdef KPEG_ABLATE_IDCT
        for (int i = 0; i < 8; ++i) {
            v[0][i] = __uint_as_float(d0.x + i) * 1e-30f;
            v[1][i] = __uint_as_float(d1.y + i) * 1e-30f;
            v[2][i] = __uint_as_float(d2.z + i) * 1e-30f;
        }
#else
        // Quantised high frequencies are mostly zero: if no block of this wavefront has a coefficient outside
        // its top-left 6x6 (luma) / 4x4 (chroma) corner, the terms of the empty rows and columns are left out
        // (same floats as the full transform, see row_idct8).  8K q75: luma 6x6 for 98 % of the tiles, chroma 4x4 for 99 %.
#if KPEG_K4_LUMA4
        // (a third luma size: 4x4 -- what smooth content quantises to)
        if (__ballot((d0.w | (u >= 6 ? (d0.x | d0.y | d0.z) : 0u)) != 0)) block_fast<8>(d0, lc, &s_m[0][u * 8], 0, v[0]);
        else if (__ballot((d0.z | (u >= 4 ? (d0.x | d0.y) : 0u)) != 0)) block_fast<6>(d0, lc, &s_m[0][u * 8], 0, v[0]);
        else block_fast<4>(d0, lc, &s_m[0][u * 8], 0, v[0]);
#else
        if (__ballot((d0.w | (u >= 6 ? (d0.x | d0.y | d0.z) : 0u)) != 0)) block_fast<8>(d0, lc, &s_m[0][u * 8], 0, v[0]);
        else block_fast<6>(d0, lc, &s_m[0][u * 8], 0, v[0]);
#endif
#if KPEG_K4_CHROMA2
        // chroma blocks of smooth content rarely have anything outside their 2x2 corner (DC and the two first-order terms)
        if (__ballot(((d1.z | d1.w) | (u >= 4 ? (d1.x | d1.y) : 0u)) != 0)) block_fast<8>(d1, lc, &s_m[1][u * 8], 1, v[1]);
        else if (__ballot((d1.y | (u >= 2 ? d1.x : 0u)) != 0)) block_fast<4>(d1, lc, &s_m[1][u * 8], 1, v[1]);
        else block_fast<2>(d1, lc, &s_m[1][u * 8], 1, v[1]);
        if (__ballot(((d2.z | d2.w) | (u >= 4 ? (d2.x | d2.y) : 0u)) != 0)) block_fast<8>(d2, lc, &s_m[1][u * 8], 1, v[2]);
        else if (__ballot((d2.y | (u >= 2 ? d2.x : 0u)) != 0)) block_fast<4>(d2, lc, &s_m[1][u * 8], 1, v[2]);
        else block_fast<2>(d2, lc, &s_m[1][u * 8], 1, v[2]);
#else
        if (__ballot(((d1.z | d1.w) | (u >= 4 ? (d1.x | d1.y) : 0u)) != 0)) block_fast<8>(d1, lc, &s_m[1][u * 8], 1, v[1]);
        else block_fast<4>(d1, lc, &s_m[1][u * 8], 1, v[1]);
        if (__ballot(((d2.z | d2.w) | (u >= 4 ? (d2.x | d2.y) : 0u)) != 0)) block_fast<8>(d2, lc, &s_m[1][u * 8], 1, v[2]);
        else block_fast<4>(d2, lc, &s_m[1][u * 8], 1, v[2]);
#endif
#endif
        // |fast - rint(fast)| + nthr >= 0  <=>  within the block's bound of a rounding boundary
        const float nthr0 = fabsf(e0) - 0.5f, nthr1 = fabsf(e1) - 0.5f, nthr2 = fabsf(e2) - 0.5f;
        // chroma samples of this MCU may exceed the f32 colour arithmetic's proven range (see block_ebound)
        const bool wide = (((__float_as_uint(e1) | __float_as_uint(e2)) & 1u) != 0) && active;
        const bool any_wide = __ballot(wide) != 0;   // wave-uniform, rare

        // Level shift + colour for the 8 pixels of this lane's row.  Per pixel one float key says
        // whether the reference-order evaluation is needed (key >= 0): a fast value within its block's
        // bound of a rounding boundary, or a G term too close to an integer for the f32 arithmetic.
        // Nearly every wavefront has a few such pixels (true ties are structural: equal and opposite
        // (0,1)/(1,0) terms cancel on a block's diagonal and leave DC/8 = n + 0.5 exactly): their positions are
        // queued straight from this loop with ballot compaction (no atomics, no second pass).
        uint32_t pk[6] = {0, 0, 0, 0, 0, 0};
        const uint32_t nq_tile = nq;
        const uint32_t pos_lane = (uint32_t)grp | ((uint32_t)lane8 << 3) | (tilek << 12) | ((__float_as_uint(e0) >> 31) << 9) |
                                  ((__float_as_uint(e1) >> 31) << 10) | ((__float_as_uint(e2) >> 31) << 11);
        const unsigned long long active_mask = __ballot(active);
        bool pushed_nc = false;   // compact stream: this tile queued a pixel with an unsafe sample in a block that is not corner-only
        uint32_t nc_lane = 0;
        constexpr int PG = COMPACT ? KPEG_PUSH_GROUP_COMPACT : KPEG_PUSH_GROUP;   // pixel columns per unsafe-pixel test
        uint32_t gsafe = 0x80000000u, sf[PG];
        // The loop exists twice: with the in-lane double colour conversion of `wide` MCUs and (nearly always) without.
        auto pixel_loop = [&](auto with_wide) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float vy = v[0][i], vb = v[1][i], vr = v[2][i];
                const float ry = __builtin_rintf(vy), rb = __builtin_rintf(vb), rr = __builtin_rintf(vr);
                // >= 0: the fast value is within its block's bound of a rounding boundary
                const float fy = fabsf(vy - ry) + nthr0, fb = fabsf(vb - rb) + nthr1, fr = fabsf(vr - rr) + nthr2;
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
                const float ka = fabsf((tc - t) - 0.5f) - (0.5f - KPEG_G_DELTA);                // >= 0: within DELTA of an integer
                const float kb = __builtin_fmaf(fabsf(t), 131072.0f, KPEG_G_DELTA - 1.0f);      // >= 0: t != 0
                uint32_t kg = __float_as_uint(ka) | __float_as_uint(kb);                         // sign clear: G is unsafe
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
                const uint32_t safe = __float_as_uint(fy) & __float_as_uint(fb) & __float_as_uint(fr) & kg;
                pk[(3 * i) >> 2] = pk_u8(R, (3 * i) & 3, pk[(3 * i) >> 2]);
                pk[(3 * i + 1) >> 2] = pk_u8(G, (3 * i + 1) & 3, pk[(3 * i + 1) >> 2]);
                pk[(3 * i + 2) >> 2] = pk_u8(B, (3 * i + 2) & 3, pk[(3 * i + 2) >> 2]);
#if defined(KPEG_ABLATE_PUSH_KEEPSAFE)
                // timing experiment: the safety arithmetic stays (kept alive), no test, no queue
                asm volatile("" ::"v"(safe));
#elif !defined(KPEG_ABLATE_PUSH)
                // A scalar branch on a vector compare costs a wavefront ~175 cycles (the VALU result has to reach the scalar
                // unit): one test per pixel column -- eight per tile, plus one more in every taken branch -- was 13 us of the
                // kernel (profiles/r02: ablations).  So: one test per GROUP of KPEG_PUSH_GROUP columns; a group with an unsafe
                // pixel (2.5 pixels per tile on the 8K workload) takes its ballots together and queues its pixels with
                // ballot compaction (no atomics).
                sf[i % PG] = safe;
                gsafe &= safe;
                if ((i % PG) == PG - 1) {
                    if (__ballot((int)gsafe >= 0) & active_mask) {
                        unsigned long long gb[PG];
#pragma unroll
                        for (int j = 0; j < PG; ++j) gb[j] = __ballot((int)sf[j] >= 0) & active_mask;
                        uint32_t base_slot = nq;
#pragma unroll
                        for (int j = 0; j < PG; ++j) {
                            const int ii = i - (PG - 1) + j;
                            const unsigned long long bal = gb[j];
                            const uint32_t slot = base_slot + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
#ifdef KPEG_ABLATE_PUSHBODY
                            if (slot == 0xFFFFFFFFu)   // timing experiment: the tests and the count stay, the entry is never written
#endif
                            if ((int)sf[j] >= 0 && active) {
                                // the pixel's rounded samples and keys again, from the fast values (cheaper than keeping them)
                                const float wy = v[0][ii], wb = v[1][ii], wr = v[2][ii];
                                const float qy = __builtin_rintf(wy), qb = __builtin_rintf(wb), qr = __builtin_rintf(wr);
                                const float gy = fabsf(wy - qy) + nthr0, gbk = fabsf(wb - qb) + nthr1, gr = fabsf(wr - qr) + nthr2;
                                const uint32_t pw = pos_lane | ((uint32_t)ii << 6);
                                if (slot < QUEUE_CAP) {
                                    uint32_t* q = s_queue + slot * QW;
                                    q[0] = pw;
                                    q[1] = __float_as_uint(qy), q[2] = __float_as_uint(qb), q[3] = __float_as_uint(qr);
                                    q[4] = __float_as_uint(gy), q[5] = __float_as_uint(gbk), q[6] = __float_as_uint(gr);
                                    // the 2x2 corner of the pixel's three blocks (rows 0 and 1, columns 0 and 1)
                                    if constexpr (COMPACT) {
                                        const uint32_t* im = s_img + (grp * 3) * 32;   // block c at + 32 c words: rows 0 and 1 at words 0 and 4
                                        q[8] = im[0], q[9] = im[4], q[10] = im[32], q[11] = im[36], q[12] = im[64], q[13] = im[68];
                                    } else if constexpr (STASH) {
                                        q[8] = cw00, q[9] = cw01, q[10] = cw10, q[11] = cw11, q[12] = cw20, q[13] = cw21;
                                    }
                                } else {
                                    s_over[slot - QUEUE_CAP] = (uint16_t)(pw & 0xFFFu);
                                }
                                if constexpr (COMPACT)
                                    nc_lane |= (~__float_as_uint(gy) & ~__float_as_uint(e0)) | (~__float_as_uint(gbk) & ~__float_as_uint(e1)) |
                                               (~__float_as_uint(gr) & ~__float_as_uint(e2));   // sign set: unsafe (key >= 0) in a block whose bound is positive (not corner-only)
                            }
                            base_slot += __popcll(bal);
                        }
                        nq = base_slot;
                    }
                    gsafe = 0x80000000u;
                }
#else
                (void)safe;
#endif
            }
        };
        if (any_wide) pixel_loop(std::true_type{});
        else pixel_loop(std::false_type{});
        if constexpr (COMPACT) pushed_nc = __ballot((int)nc_lane < 0) != 0;
        {
            // 24 bytes of pixel row lane8, MCU grp (groups beyond nm write garbage that is never stored)
            uint2* dst = reinterpret_cast<uint2*>(s_tile + lane8 * TILE_ROW_STRIDE + grp * 24);
            dst[0] = make_uint2(pk[0], pk[1]);
            dst[1] = make_uint2(pk[2], pk[3]);
            dst[2] = make_uint2(pk[4], pk[5]);
        }
        nq_total += nq - nq_tile;
        if (nq > QUEUE_CAP) {   // the rest went to the overflow list
            nover = nq - QUEUE_CAP;
            nq = QUEUE_CAP;
        }
        if constexpr (COMPACT) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#ifndef KPEG_ABLATE_RESOLVE
            if (pushed_nc) resolve_noncorner(min(nq_tile, (uint32_t)QUEUE_CAP), nq);   // while this tile's image stands
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the next iteration writes this tile back before its own colour phase overwrites the LDS tile
        }
        // Fix-ups are due: a pass's worth of queued pixels or the end of this wavefront's tiles.  Few
        // values of the tile loop are live here.  Every tile a queued position refers to must have been stored before
        // (run_fixups waits for the stores to be performed before it patches bytes of theirs).
        if (nq >= QUEUE_FLUSH || (!more && nq)) {   // wave-uniform
            if (have_prev) write_back(prev_off, prev_nm);
            have_prev = false;
            if (!p.skip_exact) run_fixups(nq, nover, tilek);
            nq = 0;
            nover = 0;
        }
        if (!more) break;
        if constexpr (PREFETCH) cur = nxt;
        tilek_cur = tilek_next;
        tilek_next = tilek_after;
        rs_next = rs_after, rn_next = rn_after;
    }
    if (have_prev) write_back(prev_off, prev_nm);
#if defined(KPEG_COUNT_COOP) || defined(KPEG_COUNT_MANY) || defined(KPEG_COUNT_FLUSH) || defined(KPEG_COUNT_CORNER)
    nq_total = dbg_count;
#else
    (void)dbg_count;
#endif
#ifdef KPEG_K4_STAMP
    if (tid == 0) {
        const unsigned long long dc = __builtin_amdgcn_s_memtime() - stamp_c0, dr = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        const uint32_t w = blockIdx.x * K4_WAVES + wave;
        if (w < 8192) {
            // every wavefront's start, end (100 MHz ticks), lifetime in shader cycles, to a slot of its own (a buffer nothing else reads)
            g_k4_stamp[w * 4 + 0] = stamp_r0;
            g_k4_stamp[w * 4 + 1] = stamp_r0 + dr;
            g_k4_stamp[w * 4 + 2] = dc;
            // unsafe pixels [63:48] | XCC_ID [35:32] | HW_ID [31:0] (wave [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13])
            g_k4_stamp[w * 4 + 3] = ((unsigned long long)(nq_total & 0xFFFFu) << 48) |
                                    ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15u) << 32) |
                                    __builtin_amdgcn_s_getreg(4 | (31 << 11));
        }
    }
#endif
    // The workgroup's last wavefront adds the workgroup's count of unsafe pixels to the statistics (spread over 256 words:
    // a single hot word serialises in L2) and takes the workgroup's end-of-call ticket.
    uint32_t last_of_wg = 0;
    if (tid == 0) {
        if (nq_total) atomicAdd(&s_wg[1], nq_total);
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

}  // namespace kpeg_dev
