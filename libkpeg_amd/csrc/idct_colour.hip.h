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
//   (2) only where the fast value lies within that bound of a rounding boundary (or an
//       operand leaves the range the fast colour arithmetic is proven for), the
//       reference-order evaluation itself, done cooperatively by one wavefront per pixel.
// Blocks with no AC coefficient are exact in (1) by construction.
//
// Mapping (no MFMA: this is HBM-bound byte/short work).
//   * 8 lanes per MCU, 8 MCUs per wavefront, 32 MCUs (256x8 pixels) per 256-thread
//     workgroup iteration; persistent grid-stride loop over tiles.
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
    uint8_t* rgb;         // output stripe base (row 0 = first pixel row of mcu_row0)
    uint32_t mcus_w;      // MCUs per MCU row (width / 8)
    uint32_t mcu_rows;    // MCU rows to produce
    uint32_t pitch;       // bytes per pixel row (width * 3)
    uint32_t tiles_w;     // ceil(mcus_w / 32)
    uint32_t ntiles;      // tiles_w * mcu_rows
    uint32_t* stats;      // [0] += pixels sent to the exact path (may be null)
};

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

// One sample of MCU::computeIDCT evaluated by a whole wavefront: lane p owns coefficient
// position p = u*8+v (row-major = the reference's loop order), computes its product
// term; the float accumulation then walks the non-zero lanes in order.
// F: this lane's dequantised coefficient (int), x = pixel row, y = pixel column.
// Returns S = (int)roundl(ic) + 128 in every lane.
__device__ __forceinline__ int exact_sample_wave(int F, int x, int y)
{
    const int lane = __lane_id();
    const int u = lane >> 3, v = lane & 7;
    float fc = cc_of(u, v) * (float)F;                                 // float multiply
    double t = ((double)fc * c_cos[x * 8 + u]) * c_cos[y * 8 + v];      // two double multiplies
    unsigned long long live = __ballot(F != 0);
    float sum = 0.0f;
    while (live) {
        int p = __builtin_ctzll(live);
        live &= live - 1;
        // p is wave-uniform: two v_readlane_b32
        long long tb = __builtin_bit_cast(long long, t);
        unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)tb, p);
        unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(tb >> 32), p);
        double tp = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        sum = (float)((double)sum + tp);
    }
    float ic = (float)(0.25 * (double)sum);
    return level_shift(ic);
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

// ---- mode 0: fast path + exact re-evaluation ---------------------------------------------

// Error-bound constants; derivation and numeric check: tools/idct_bound.py, DESIGN.md.
//   |fast - reference float result| <= U * A * (nnz_bound + KAPPA)     (sample units)
// with U = 2^-24 (1 + 2^-20), A = sum |in| over the block, when the block has AC terms.
#define KPEG_KAPPA 24.0f
#define KPEG_U 0x1.00001p-24f
// chroma magnitude below which the f32 colour arithmetic is proven exact
#define KPEG_CHROMA_LIM 250.0f
#define KPEG_LUMA_LIM 1048576.0f
// |t - rint(t)| below this sends the G channel to the exact path (f32 error of t <= 3.7e-5)
#define KPEG_G_DELTA 6.0e-5f

constexpr int TILE_MCUS = 32;                 // MCUs per workgroup iteration
constexpr int TILE_ROW_BYTES = TILE_MCUS * 24;  // 768
constexpr int TILE_ROW_STRIDE = 816;          // padded: 204 dwords = 12 mod 32 banks
constexpr int QUEUE_CAP = TILE_MCUS * 64;     // every pixel of the tile

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dppi(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
#define DPP_QUAD_BCAST(k) ((k) | ((k) << 2) | ((k) << 4) | ((k) << 6))
#define DPP_QUAD_XOR1 0xB1   // [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E   // [2,3,0,1]
#define DPP_HALF_MIRROR 0x141

// 1-D 8-point inverse DCT kernel sum_v a[v] cos((2y+1) v pi/16), y = 0..7, in place.
__device__ __forceinline__ void row_idct8(float a[8])
{
    const float c1 = 0.98078528040323044913f, c2 = 0.92387953251128675613f, c3 = 0.83146961230254523708f,
                c4 = 0.70710678118654752440f, c5 = 0.55557023301960222474f, c6 = 0.38268343236508977173f,
                c7 = 0.19509032201612826785f;
    float t0 = __builtin_fmaf(a[4], c4, a[0]);
    float t1 = __builtin_fmaf(a[4], -c4, a[0]);
    float p = __builtin_fmaf(a[6], c6, a[2] * c2);
    float q = __builtin_fmaf(a[6], -c2, a[2] * c6);
    float e0 = t0 + p, e3 = t0 - p, e1 = t1 + q, e2 = t1 - q;
    float o0 = __builtin_fmaf(a[7], c7, __builtin_fmaf(a[5], c5, __builtin_fmaf(a[3], c3, a[1] * c1)));
    float o1 = __builtin_fmaf(a[7], -c5, __builtin_fmaf(a[5], -c1, __builtin_fmaf(a[3], -c7, a[1] * c3)));
    float o2 = __builtin_fmaf(a[7], c3, __builtin_fmaf(a[5], c7, __builtin_fmaf(a[3], -c1, a[1] * c5)));
    float o3 = __builtin_fmaf(a[7], -c1, __builtin_fmaf(a[5], c3, __builtin_fmaf(a[3], -c5, a[1] * c7)));
    a[0] = e0 + o0;
    a[7] = e0 - o0;
    a[1] = e1 + o1;
    a[6] = e1 - o1;
    a[2] = e2 + o2;
    a[5] = e2 - o2;
    a[3] = e3 + o3;
    a[4] = e3 - o3;
}

__device__ __forceinline__ float cosf_tab(int k)  // cos(k*pi/16), k = 0..31, f32-rounded
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
    float m[2][8];   // AC input scale 0.25 * cc[u][v] * Q[u][v] for Y / chroma tables
    float q0[2];     // Q[u][0] as float (exact DC-column chain)
    float cc0;       // cc[u][0]
    float k[4];      // column-pass constants
    float s;         // -1 on even-row lanes, +1 on odd-row lanes
    uint32_t dcmask; // clears the DC halfword on the lane that owns row 0
};

// One component block: loads are already in d[4] (8 int16, row u of the block).
// Returns the 8 fast sample values of pixel row (lane & 7) in out[8] and the
// block's unsafe threshold (0.5 - E).
__device__ __forceinline__ float block_fast(const uint4 d, const LaneConst& lc, int tab, float out[8])
{
    int w[4] = {(int)d.x, (int)d.y, (int)d.z, (int)d.w};
    float a[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[2 * i] = (float)(short)(w[i] & 0xFFFF);
        a[2 * i + 1] = (float)(w[i] >> 16);
    }
    // nnz bound: sum of squares of the AC coefficients (>= number of non-zero ones),
    // saturating, then clamped per lane so that the cross-lane sum cannot overflow
    int n = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, (int)(w[0] & lc.dcmask)),
                                   __builtin_bit_cast(short2v, (int)(w[0] & lc.dcmask)), 0, true);
#pragma unroll
    for (int i = 1; i < 4; ++i)
        n = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, w[i]), __builtin_bit_cast(short2v, w[i]), n, true);
    n = min(n, 63);
    // inputs: column 0 through the reference's own chain 0.25 * (cc * (float)(c*Q)), exact for DC
    float in0 = 0.25f * (lc.cc0 * (a[0] * lc.q0[tab]));
    a[0] = in0;
    float A = fabsf(in0);
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        a[i] *= lc.m[tab][i];
        A += fabsf(a[i]);
    }
    row_idct8(a);
    // block totals over the 8 lanes of the group
    A += dpp<DPP_HALF_MIRROR>(A);
    A += dpp<DPP_QUAD_XOR1>(A);
    A += dpp<DPP_QUAD_XOR2>(A);
    n += dppi<DPP_HALF_MIRROR>(n);
    n += dppi<DPP_QUAD_XOR1>(n);
    n += dppi<DPP_QUAD_XOR2>(n);
    float nb = (float)min(n, 63);
    float E = (KPEG_U * A) * (nb + (n > 0 ? KPEG_KAPPA : 0.0f));
    // column pass across lanes
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        float g = a[y];
        float acc = dpp<DPP_QUAD_BCAST(0)>(g) * lc.k[0];
        acc = __builtin_fmaf(dpp<DPP_QUAD_BCAST(1)>(g), lc.k[1], acc);
        acc = __builtin_fmaf(dpp<DPP_QUAD_BCAST(2)>(g), lc.k[2], acc);
        acc = __builtin_fmaf(dpp<DPP_QUAD_BCAST(3)>(g), lc.k[3], acc);
        out[y] = __builtin_fmaf(dpp<DPP_HALF_MIRROR>(acc), lc.s, acc);
    }
    return 0.5f - E;
}

__global__ __launch_bounds__(256) void k_idct_colour_fast(IdctParams p, QTables qt)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[8 * TILE_ROW_STRIDE];
    __shared__ uint32_t s_queue[QUEUE_CAP];
    __shared__ uint32_t s_qcount;

    const int tid = threadIdx.x;
    const int lane8 = tid & 7;          // lane within the MCU group = output pixel row
    const int grp = tid >> 3;           // MCU within the tile, 0..31
    const int u = lane8 < 4 ? 2 * lane8 : 2 * (lane8 - 4) + 1;  // coefficient row this lane loads

    LaneConst lc;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int v = 0; v < 8; ++v) lc.m[t][v] = 0.25f * cc_of(u, v) * (float)qt.q[t][u * 8 + v];
        lc.q0[t] = (float)qt.q[t][u * 8];
    }
    lc.cc0 = cc_of(u, 0);
    lc.dcmask = (u == 0) ? 0xFFFF0000u : 0xFFFFFFFFu;
    if (lane8 < 4) {
        // even rows 0,2,4,6 live on lanes 0..3; this lane produces E_x, x = lane8
#pragma unroll
        for (int k = 0; k < 4; ++k) lc.k[k] = cosf_tab((2 * lane8 + 1) * (2 * k));
        lc.s = 1.0f;
    } else {
        // odd rows 1,3,5,7 live on lanes 4..7; this lane produces -O_x, x = 7 - lane8
        const int x = 7 - lane8;
#pragma unroll
        for (int k = 0; k < 4; ++k) lc.k[k] = -cosf_tab((2 * x + 1) * (2 * k + 1));
        lc.s = 1.0f;
    }
    // combine: out = own + mirror * s.  Even lane x: E_x + (-(-O_x))  -> s = -1 on even lanes;
    // odd lane (row 7-x): E_x - O_x = mirror(E_x) * 1 + own(-O_x)   -> s = +1 on odd lanes.
    lc.s = lane8 < 4 ? -1.0f : 1.0f;

    for (uint32_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const uint32_t trow = tile / p.tiles_w, tcol = tile - trow * p.tiles_w;
        const uint32_t m0 = tcol * TILE_MCUS;                       // first MCU column of the tile
        const uint32_t nm = min((uint32_t)TILE_MCUS, p.mcus_w - m0);  // MCUs in this tile
        const bool active = (uint32_t)grp < nm;

        if (tid == 0) s_qcount = 0;

        float v[3][8];
        float thr[3];
        {
            const size_t mcu = (size_t)trow * p.mcus_w + m0 + (active ? grp : 0);
            const uint4* src = reinterpret_cast<const uint4*>(p.coef) + mcu * 24 + u;
            uint4 d0 = src[0], d1 = src[8], d2 = src[16];
            thr[0] = block_fast(d0, lc, 0, v[0]);
            thr[1] = block_fast(d1, lc, 1, v[1]);
            thr[2] = block_fast(d2, lc, 1, v[2]);
        }
        __syncthreads();  // s_qcount reset visible; previous iteration's tile reads done

        // level shift + colour for the 8 pixels of this lane's row
        float fmaxv[3] = {0.f, 0.f, 0.f};
        float amaxY = 0.f, amaxC = 0.f, gmin = 1.0f;
        uint32_t packed[6];
        uint32_t bytes[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float ry = __builtin_rintf(v[0][i]), rb = __builtin_rintf(v[1][i]), rr = __builtin_rintf(v[2][i]);
            fmaxv[0] = fmaxf(fmaxv[0], fabsf(v[0][i] - ry));
            fmaxv[1] = fmaxf(fmaxv[1], fabsf(v[1][i] - rb));
            fmaxv[2] = fmaxf(fmaxv[2], fabsf(v[2][i] - rr));
            amaxY = fmaxf(amaxY, fabsf(v[0][i]));
            amaxC = fmaxf(amaxC, fmaxf(fabsf(v[1][i]), fabsf(v[2][i])));
            float yf = ry + 128.0f;
            float R = yf + floorf(rr * 1.402f);
            float B = yf + floorf(rb * 1.772f);
            float t = __builtin_fmaf(rr, 0.714136f, rb * 0.344136f);
            float G = yf - ceilf(t);
            float dt = fabsf(t - __builtin_rintf(t));
            dt = (t == 0.0f) ? 1.0f : dt;
            gmin = fminf(gmin, dt);
            int Ri = (int)fminf(fmaxf(R, 0.f), 255.f);
            int Gi = (int)fminf(fmaxf(G, 0.f), 255.f);
            int Bi = (int)fminf(fmaxf(B, 0.f), 255.f);
            bytes[i] = (uint32_t)Ri | ((uint32_t)Gi << 8) | ((uint32_t)Bi << 16);
        }
        // 8 x 3 bytes -> 6 dwords
        packed[0] = bytes[0] | (bytes[1] << 24);
        packed[1] = (bytes[1] >> 8) | (bytes[2] << 16);
        packed[2] = (bytes[2] >> 16) | (bytes[3] << 8);
        packed[3] = bytes[4] | (bytes[5] << 24);
        packed[4] = (bytes[5] >> 8) | (bytes[6] << 16);
        packed[5] = (bytes[6] >> 16) | (bytes[7] << 8);

        if (active) {
            uint2* dst = reinterpret_cast<uint2*>(s_tile + lane8 * TILE_ROW_STRIDE + grp * 24);
            dst[0] = make_uint2(packed[0], packed[1]);
            dst[1] = make_uint2(packed[2], packed[3]);
            dst[2] = make_uint2(packed[4], packed[5]);
        }

        const bool suspicious = (fmaxv[0] >= thr[0]) | (fmaxv[1] >= thr[1]) | (fmaxv[2] >= thr[2]) |
                                (amaxY >= KPEG_LUMA_LIM) | (amaxC >= KPEG_CHROMA_LIM) | (gmin < KPEG_G_DELTA);
        if (suspicious && active) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float ry = __builtin_rintf(v[0][i]), rb = __builtin_rintf(v[1][i]), rr = __builtin_rintf(v[2][i]);
                uint32_t mask = 0;
                if (fabsf(v[0][i] - ry) >= thr[0] || fabsf(v[0][i]) >= KPEG_LUMA_LIM) mask |= 1;
                if (fabsf(v[1][i] - rb) >= thr[1] || fabsf(v[1][i]) >= KPEG_CHROMA_LIM) mask |= 2;
                if (fabsf(v[2][i] - rr) >= thr[2] || fabsf(v[2][i]) >= KPEG_CHROMA_LIM) mask |= 4;
                float t = __builtin_fmaf(rr, 0.714136f, rb * 0.344136f);
                float dt = fabsf(t - __builtin_rintf(t));
                bool colour = (t != 0.0f) && (dt < KPEG_G_DELTA);
                if (mask || colour) {
                    uint32_t slot = atomicAdd(&s_qcount, 1u);
                    // entry: [4:0] mcu in tile, [7:5] row, [10:8] col, [13:11] comps to re-evaluate
                    s_queue[slot] = (uint32_t)grp | ((uint32_t)lane8 << 5) | ((uint32_t)i << 8) | (mask << 11);
                }
            }
        }
        __syncthreads();

        // exact re-evaluation, one wavefront per queued pixel
        const uint32_t nq = s_qcount;
        if (nq) {
            const int wave = tid >> 6, lane = tid & 63;
            for (uint32_t e = wave; e < nq; e += 4) {
                const uint32_t ent = s_queue[e];
                const int g = ent & 31, x = (ent >> 5) & 7, y = (ent >> 8) & 7;
                const size_t mcu = (size_t)trow * p.mcus_w + m0 + g;
                int S[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // every component is re-evaluated: the fast value of an unflagged one is
                    // provably the same, and the pixel needs all three as integers anyway
                    int F = (int)p.coef[(mcu * 3 + c) * 64 + lane] * (int)qt.q[c ? 1 : 0][lane];
                    S[c] = exact_sample_wave(F, x, y);
                }
                if (lane == 0) {
                    uint32_t px = colour_exact(S[0], S[1], S[2]);
                    uint8_t* o = s_tile + x * TILE_ROW_STRIDE + g * 24 + y * 3;
                    o[0] = (uint8_t)px;
                    o[1] = (uint8_t)(px >> 8);
                    o[2] = (uint8_t)(px >> 16);
                }
            }
            if (tid == 0 && p.stats) atomicAdd(p.stats, nq);
            __syncthreads();
        }

        // coalesced write-back of the tile: 8 rows x nm*24 bytes
        {
            uint8_t* base = p.rgb + (size_t)trow * 8 * p.pitch + (size_t)m0 * 24;
            const uint32_t row_bytes = nm * 24;
            if (nm == TILE_MCUS && ((reinterpret_cast<uintptr_t>(base) | p.pitch) & 15) == 0) {
                for (int c = tid; c < 8 * (TILE_ROW_BYTES / 16); c += 256) {
                    int r = c / (TILE_ROW_BYTES / 16), k = c - r * (TILE_ROW_BYTES / 16);
                    uint4 val = *reinterpret_cast<const uint4*>(s_tile + r * TILE_ROW_STRIDE + k * 16);
                    *reinterpret_cast<uint4*>(base + (size_t)r * p.pitch + k * 16) = val;
                }
            } else {
                const uint32_t per_row = row_bytes / 8;
                for (uint32_t c = tid; c < 8 * per_row; c += 256) {
                    uint32_t r = c / per_row, k = c - r * per_row;
                    uint2 val = *reinterpret_cast<const uint2*>(s_tile + r * TILE_ROW_STRIDE + k * 8);
                    *reinterpret_cast<uint2*>(base + (size_t)r * p.pitch + k * 8) = val;
                }
            }
        }
        // the next iteration's first barrier orders these LDS reads before the next writes
    }
}

}  // namespace kpeg_dev
