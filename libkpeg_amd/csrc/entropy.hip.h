// libkpeg_amd/csrc/entropy.hip.h -- K0..K2: on-device entropy decode for gfx950.
//
// Replaces JPEGDecoder::byteStuffScanData (src/Decoder.cpp:621-653), the Huffman bit loop
// of JPEGDecoder::decodeScanData (:694-803) with HuffmanTree::contains
// (src/HuffmanTree.cpp:164-193) and bitStringtoValue (src/Image.cpp:285-302), and the RLE
// walk + DC prediction of MCU::constructMCU (src/MCU.cpp:91-108) of the reference.
//
// A baseline scan without restart markers is one serial bit string, so the decode is the
// two-pass "self-synchronising sub-sequence" scheme:
//   K0  unstuff   byte-parallel removal of the 00 after FF (reference rule incl. its tail
//                 rule) and, with DRI, of the RSTn markers; prefix-sum compaction.  Output
//                 is the bit string as big-endian 32-bit words + restart-segment offsets.
//   K1  sync      one lane per sub-sequence of SUBSEQ_BITS bits: decode from a guessed
//                 codeword boundary, then Jacobi rounds "re-decode from my predecessor's
//                 exit state until nobody's exit state changes".  Huffman streams
//                 re-synchronise after a few symbols, so a handful of rounds suffice; the
//                 first sub-sequence of every restart segment starts from a known state.
//                 Each run also counts the blocks it starts and sums their DC differences.
//       scan      exclusive prefix sum of (blocks, dc[3]) over sub-sequences: absolute
//                 block index and DC predictors at every sub-sequence entry.
//   K2  write     one lane per sub-sequence decodes again from its true entry state and
//                 writes whole 128-byte coefficient blocks (natural order, absolute DC,
//                 quirk Q1 applied) -- the layout K4 reads.
// Huffman look-up tables (9-bit first level + canonical long-code search) live in LDS.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <string>

#include "../../include/kpeg_hip.h"
#include "kpeg_tables.h"

namespace kpeg_dev {

#ifndef KPEG_ABLATE_W
#define KPEG_ABLATE_W 0   // timing experiments only (tools/variants.sh)
#endif
#ifndef KPEG_SYNC_STATS
#define KPEG_SYNC_STATS 0
#endif
#ifndef KPEG_SUBSEQ_BITS
#define KPEG_SUBSEQ_BITS 256
#endif
constexpr int SUBSEQ_BITS = KPEG_SUBSEQ_BITS;  // bits per sub-sequence (tunable; 128, 256, 512 or 1024)
constexpr int SUBSEQ_WORDS = SUBSEQ_BITS / 32;
constexpr int LUT_BITS = 9;
constexpr int SYNC_PASSES = 4;  // sync kernels enqueued per call: pass 0 + boundary passes (idle ones exit at once)
constexpr int SYNC_WG = 256;    // sub-sequences per workgroup

constexpr int LUT2_BITS = 16 - LUT_BITS;   // remaining bits of a long code
constexpr int LUT2_SUBS = 12;             // second-level tables per Huffman table (canonical codes need few)
constexpr uint16_t LUT_LONG = 0x8000;     // first-level entry: code longer than LUT_BITS, low bits = sub-table
constexpr uint16_t LUT_SEARCH = 0xFFFF;   // ... no sub-table left: canonical search (never for real tables)

struct EntropyTables {  // built on the host per frame, copied to the device when it changes
    uint16_t lut[4][1 << LUT_BITS];  // [class*2+id]: (len << 8) | symbol; LUT_LONG | sub; 0 = no such code
    uint16_t lut2[4][LUT2_SUBS][1 << LUT2_BITS];  // (len << 8) | symbol for codes of 10..16 bits; 0 = no such code
    int32_t maxcode[4][18];          // canonical: largest code of each length (-1 if none)
    int32_t valoff[4][18];           // symbol index of the first code of each length minus that code
    uint8_t symbols[4][256];
    uint8_t zz[64];                  // zig-zag -> natural
    float mscale[2][64];             // 0.25 * cc[u][v] * Q[u][v], natural order ([.][0] unused): K4's input scale
    float q00[2];                    // Q[0][0] of both tables
};

static int build_entropy_tables(const kpeg_frame* f, EntropyTables* t)
{
    std::memset(t, 0, sizeof(*t));
    for (int k = 0; k < 64; ++k) t->zz[k] = KPEG_ZZ_TO_NATURAL[k];
    for (int tq = 0; tq < 2; ++tq) {
        const float c0 = 0x1.6a09e6p-1f;  // (float)(1/sqrt 2), as in K4's cc_of()
        for (int k = 0; k < 64; ++k) {
            const int nat = KPEG_ZZ_TO_NATURAL[k], u = nat >> 3, v = nat & 7;
            const float cc = (u == 0 ? c0 : 1.0f) * (v == 0 ? c0 : 1.0f);
            t->mscale[tq][nat] = 0.25f * cc * (float)f->qt[tq][k];
        }
        t->q00[tq] = (float)f->qt[tq][0];
    }
    for (int cls = 0; cls < 2; ++cls)
        for (int id = 0; id < 2; ++id) {
            const kpeg_dht& h = f->dht[cls][id];
            const int ti = cls * 2 + id;
            int code = 0, k = 0, nsub = 0;
            for (int len = 1; len <= 16; ++len) {
                int cnt = h.counts[len - 1];
                if (k + cnt > 256) return -1;
                t->valoff[ti][len] = k - code;
                t->maxcode[ti][len] = cnt ? code + cnt - 1 : -1;
                if (cnt && code + cnt - 1 >= (1 << len)) return -1;  // not a prefix code
                for (int i = 0; i < cnt; ++i) {
                    uint8_t sym = h.symbols[k];
                    t->symbols[ti][k] = sym;
                    const uint16_t e = (uint16_t)((len << 8) | sym);
                    if (len <= LUT_BITS) {
                        int first = code << (LUT_BITS - len);
                        for (int j = 0; j < (1 << (LUT_BITS - len)); ++j) t->lut[ti][first + j] = e;
                    } else {
                        const int prefix = code >> (len - LUT_BITS);
                        uint16_t& l1 = t->lut[ti][prefix];
                        if (l1 == 0) l1 = nsub < LUT2_SUBS ? (uint16_t)(LUT_LONG | nsub++) : LUT_SEARCH;
                        if (l1 != LUT_SEARCH) {
                            const int sub = l1 & 0x7FFF, rem = len - LUT_BITS;
                            const int first = (code & ((1 << rem) - 1)) << (LUT2_BITS - rem);
                            for (int j = 0; j < (1 << (LUT2_BITS - rem)); ++j) t->lut2[ti][sub][first + j] = e;
                        }
                    }
                    code++;
                    k++;
                }
                code <<= 1;
            }
            if (k == 0) return -1;
        }
    return 0;
}

struct EntropyMeta {   // device-resident bookkeeping written by K0
    uint32_t n_u;      // un-stuffed length in bytes
    uint32_t nseg;     // restart segments found (markers + 1)
    uint32_t nsub;     // total sub-sequences
    uint32_t moved[SYNC_PASSES + 8];  // per pass: workgroups whose last exit state moved
    uint32_t total_blocks;
};

struct EntropyScratch {
    void* d_u = nullptr;        size_t u_cap = 0;       // un-stuffed words
    void* d_part = nullptr;     size_t part_cap = 0;    // per-block partial sums of K0
    void* d_segoff = nullptr;   size_t seg_cap = 0;     // seg_off[S+1], sub_base[S+1]
    void* d_state = nullptr;    size_t state_cap = 0;   // X[nsub] uint64, Xb[2][nwg] uint64, mv[2][nwg] uint8
    void* d_cnt = nullptr;      size_t cnt_cap = 0;     // cnt[nsub] int4, prefix[nsub] int4
    void* d_wsum = nullptr;     size_t wsum_cap = 0;
    EntropyMeta* d_meta = nullptr;
    EntropyTables* d_tabs = nullptr;
    EntropyTables h_tabs_cached;
    bool tabs_valid = false;
};

static void entropy_scratch_free(EntropyScratch* s)
{
    void* ps[] = {s->d_u, s->d_part, s->d_segoff, s->d_state, s->d_cnt, s->d_wsum, s->d_meta, s->d_tabs};
    for (void* p : ps)
        if (p) (void)hipFree(p);
    *s = EntropyScratch();
}

struct EntropyLaunch {
    hipStream_t stream;
    const uint8_t* d_scan;
    size_t scan_len;
    uint32_t nmcu;
    uint32_t restart_interval;
    int16_t* d_coef;
    float* d_ebound;
    uint32_t* d_status;
    int num_cus;
    int sync_passes;   // 0 = default
};

// ------------------------------------------------------------------------------------------
// K0: unstuff
// keep(j): reference stream: drop b[j]==00 after b[j-1]==FF unless j is the last byte
// (byteStuffScanData's `i + 8 < size - 8`, Decoder.cpp:637).  With restart markers: drop every
// stuffed 00 and both bytes of FF D0..D7.
constexpr int US_BYTES_PER_THREAD = 16;
constexpr int US_THREADS = 256;
constexpr int US_BLOCK_BYTES = US_BYTES_PER_THREAD * US_THREADS;

__device__ __forceinline__ void us_flags(const uint8_t* b, uint32_t n, uint32_t j0, bool rst, uint32_t& keepmask,
                                         uint32_t& markmask)
{
    keepmask = 0;
    markmask = 0;
    uint32_t prev = j0 > 0 && j0 - 1 < n ? b[j0 - 1] : 0x100;
    for (int k = 0; k < US_BYTES_PER_THREAD; ++k) {
        uint32_t j = j0 + k;
        if (j >= n) break;
        uint32_t cur = b[j];
        bool keep = true;
        if (rst) {
            uint32_t nxt = j + 1 < n ? b[j + 1] : 0x100;
            if (cur == 0x00 && prev == 0xFF) keep = false;
            if (cur == 0xFF && nxt >= 0xD0 && nxt <= 0xD7) {
                keep = false;
                markmask |= 1u << k;  // a marker starts here
            }
            if (prev == 0xFF && cur >= 0xD0 && cur <= 0xD7) keep = false;
        } else {
            if (cur == 0x00 && prev == 0xFF && j + 1 < n) keep = false;  // never the last byte
        }
        if (keep) keepmask |= 1u << k;
        prev = cur;
    }
}

__global__ __launch_bounds__(US_THREADS) void k_unstuff_count(const uint8_t* b, uint32_t n, int rst, uint2* part)
{
    __shared__ uint32_t s_k[US_THREADS / 64], s_m[US_THREADS / 64];
    uint32_t j0 = (blockIdx.x * US_THREADS + threadIdx.x) * US_BYTES_PER_THREAD;
    uint32_t km, mm;
    us_flags(b, n, j0, rst != 0, km, mm);
    uint32_t k = __popc(km), m = __popc(mm);
    for (int o = 32; o > 0; o >>= 1) {
        k += __shfl_down(k, o);
        m += __shfl_down(m, o);
    }
    if ((threadIdx.x & 63) == 0) {
        s_k[threadIdx.x >> 6] = k;
        s_m[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tk = 0, tm = 0;
        for (int i = 0; i < US_THREADS / 64; ++i) {
            tk += s_k[i];
            tm += s_m[i];
        }
        part[blockIdx.x] = make_uint2(tk, tm);
    }
}

// single workgroup: exclusive scan of part[] in place, totals to meta
__global__ __launch_bounds__(1024) void k_unstuff_scan(uint2* part, uint32_t nparts, EntropyMeta* meta)
{
    __shared__ uint2 s[1024];
    __shared__ uint2 carry;
    if (threadIdx.x == 0) carry = make_uint2(0, 0);
    __syncthreads();
    for (uint32_t base = 0; base < nparts; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint2 v = i < nparts ? part[i] : make_uint2(0, 0);
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint2 t = make_uint2(0, 0);
            if ((int)threadIdx.x >= o) t = s[threadIdx.x - o];
            __syncthreads();
            s[threadIdx.x].x += t.x;
            s[threadIdx.x].y += t.y;
            __syncthreads();
        }
        uint2 incl = s[threadIdx.x];
        uint2 c = carry;
        if (i < nparts) part[i] = make_uint2(c.x + incl.x - v.x, c.y + incl.y - v.y);
        __syncthreads();
        if (threadIdx.x == 1023) carry = make_uint2(c.x + incl.x, c.y + incl.y);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        meta->n_u = carry.x;
        meta->nseg = carry.y + 1;
    }
}

__global__ __launch_bounds__(US_THREADS) void k_unstuff_scatter(const uint8_t* b, uint32_t n, int rst, const uint2* part,
                                                                uint8_t* u, uint32_t* seg_off, uint32_t seg_cap)
{
    __shared__ uint32_t s_k[US_THREADS], s_m[US_THREADS];
    uint32_t j0 = (blockIdx.x * US_THREADS + threadIdx.x) * US_BYTES_PER_THREAD;
    uint32_t km, mm;
    us_flags(b, n, j0, rst != 0, km, mm);
    s_k[threadIdx.x] = __popc(km);
    s_m[threadIdx.x] = __popc(mm);
    __syncthreads();
    // simple Hillis-Steele inclusive scan over 256 entries
    for (int o = 1; o < US_THREADS; o <<= 1) {
        uint32_t a = 0, c = 0;
        if ((int)threadIdx.x >= o) {
            a = s_k[threadIdx.x - o];
            c = s_m[threadIdx.x - o];
        }
        __syncthreads();
        s_k[threadIdx.x] += a;
        s_m[threadIdx.x] += c;
        __syncthreads();
    }
    uint2 base = part[blockIdx.x];
    uint32_t pos = base.x + s_k[threadIdx.x] - __popc(km);
    uint32_t mk = base.y + s_m[threadIdx.x] - __popc(mm);
    for (int k = 0; k < US_BYTES_PER_THREAD; ++k) {
        uint32_t j = j0 + k;
        if (j >= n) break;
        if (mm & (1u << k)) {
            mk++;
            if (mk < seg_cap) seg_off[mk] = pos;  // segment mk starts at the next kept byte
        }
        if (km & (1u << k)) {
            u[pos ^ 3] = b[j];  // big-endian words for little-endian 32-bit loads
            pos++;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) seg_off[0] = 0;
}

// single workgroup: seg_off[nseg] = n_u, zero the padding words, sub_base[] = exclusive scan
// of ceil(seg_bits / SUBSEQ_BITS)
__global__ __launch_bounds__(1024) void k_seg_setup(EntropyMeta* meta, uint32_t* seg_off, uint32_t* sub_base, uint32_t seg_cap,
                                                    uint32_t nsub_cap, uint8_t* u, uint32_t expected_segs, uint32_t* status)
{
    __shared__ uint32_t s[1024];
    __shared__ uint32_t carry;
    const uint32_t n_u = meta->n_u;
    uint32_t nseg = meta->nseg;
    if (nseg != expected_segs || nseg + 1 > seg_cap) {
        if (threadIdx.x == 0) {
            atomicOr(&status[1], 1u);  // restart markers do not match the restart interval
            meta->nsub = 0;
        }
        return;
    }
    if (threadIdx.x == 0) {
        seg_off[nseg] = n_u;
        carry = 0;
    }
    // bytes past the end read as zero (the reader may look 8 bytes ahead)
    if (threadIdx.x < 16) {
        uint32_t j = (n_u + threadIdx.x);
        // only the bytes beyond n_u inside the last partially written word and two more words
        u[j ^ 3] = 0;
    }
    __syncthreads();
    for (uint32_t base = 0; base < nseg; base += 1024) {
        uint32_t r = base + threadIdx.x;
        uint32_t v = 0;
        if (r < nseg) {
            uint32_t bits = (seg_off[r + 1] - seg_off[r]) * 8;
            v = (bits + SUBSEQ_BITS - 1) / SUBSEQ_BITS;
            if (v == 0) v = 1;
        }
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint32_t t = 0;
            if ((int)threadIdx.x >= o) t = s[threadIdx.x - o];
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t c = carry;
        if (r < nseg) sub_base[r] = c + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sub_base[nseg] = carry;
        if (carry > nsub_cap) {
            atomicOr(&status[1], 2u);
            meta->nsub = 0;
        } else {
            meta->nsub = carry;
        }
        for (int i = 0; i < SYNC_PASSES + 8; ++i) meta->moved[i] = 0;
        meta->total_blocks = 0;
    }
}

// ------------------------------------------------------------------------------------------
// Decoder core

struct DecState {   // at a codeword boundary
    uint32_t p;     // bit position in the un-stuffed string
    uint32_t c;     // component 0..2 of the block being decoded
    uint32_t k;     // 0: next symbol is the DC symbol; 1..63: AC, k-1 coefficients placed so far
};
__device__ __forceinline__ uint64_t pack_state(const DecState& s)
{
    return (uint64_t)s.p | ((uint64_t)s.c << 32) | ((uint64_t)s.k << 34);
}
__device__ __forceinline__ DecState unpack_state(uint64_t v)
{
    DecState s;
    s.p = (uint32_t)v;
    s.c = (uint32_t)(v >> 32) & 3;
    s.k = (uint32_t)(v >> 34) & 127;
    return s;
}

struct LdsTables {
    uint16_t lut[4][1 << LUT_BITS];
    uint16_t lut2[4][LUT2_SUBS][1 << LUT2_BITS];
    int32_t maxcode[4][18];
    int32_t valoff[4][18];
    uint8_t symbols[4][256];
    uint8_t zz[64];
    float mscale[2][64];
    float q00[2];
};

__device__ __forceinline__ void load_tables(LdsTables* dst, const EntropyTables* src)
{
    static_assert(sizeof(LdsTables) == sizeof(EntropyTables), "layout");
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    for (uint32_t i = threadIdx.x; i < sizeof(LdsTables) / 4; i += blockDim.x) d[i] = s[i];
}

// The bit string as the decode loops see it: the workgroup's slice staged in LDS (a symbol costs
// two LDS reads instead of two dependent global loads), anything beyond the slice from global memory.
// LDS word j lives at j + j/SUBSEQ_WORDS: lanes start SUBSEQ_WORDS words apart, the padding spreads
// them over the banks.
constexpr int STAGE_WORDS = SYNC_WG * (SUBSEQ_BITS / 32);   // the workgroup's own sub-sequences
constexpr int STAGE_MARGIN = 512;                           // words of run-over into the next workgroup
constexpr int STAGE_TOTAL = STAGE_WORDS + STAGE_MARGIN;
constexpr int STAGE_LDS = STAGE_TOTAL + STAGE_TOTAL / SUBSEQ_WORDS + 2;
__device__ __forceinline__ uint32_t stage_idx(uint32_t j) { return j + j / SUBSEQ_WORDS; }

struct BitSrc {
    const uint32_t* __restrict__ g;   // whole un-stuffed string
    const uint32_t* lds;              // staged slice
    uint32_t w0;                      // first staged word
    uint32_t nw;                      // staged words
};

__device__ __forceinline__ void stage_bits(uint32_t* lds, const uint32_t* __restrict__ g, uint32_t w0, uint32_t nw)
{
    for (uint32_t j = threadIdx.x; j < nw; j += blockDim.x) lds[stage_idx(j)] = g[w0 + j];
}

// Word i of the bit string for the decode loops.  Every position a lane can reach lies inside its
// workgroup's staged slice: a run starts at most one symbol (< 32 bits) before its own sub-sequence
// and K2 finishes at most one block (< 1728 bits, STAGE_MARGIN is 16384) past it.  Only a corrupt
// stream can run further; its reads are clamped (the result is flagged as an error elsewhere).
__device__ __forceinline__ uint32_t src_word(const BitSrc& b, uint32_t i)
{
    const uint32_t j = min(i - b.w0, b.nw - 1);
    return b.lds[stage_idx(j)];
}

// Sequential reader: the next >= 32 bits sit MSB-aligned in a 64-bit register, so a symbol's
// critical path is one LUT read, not LUT + two word fetches; the following word is fetched one
// refill ahead.
struct BitReader {
    uint64_t buf;
    uint32_t have;   // valid bits in buf, >= 32 between symbols
    uint32_t wi;     // index of the word after `nextw`
    uint32_t nextw;

    __device__ __forceinline__ void init(const BitSrc& b, uint32_t p)
    {
        const uint32_t i = p >> 5, o = p & 31;
        buf = (((uint64_t)src_word(b, i) << 32) | src_word(b, i + 1)) << o;
        have = 64 - o;
        nextw = src_word(b, i + 2);
        wi = i + 3;
    }
    __device__ __forceinline__ uint32_t peek() const { return (uint32_t)(buf >> 32); }
    __device__ __forceinline__ void consume(const BitSrc& b, uint32_t n)   // n <= 31
    {
        buf <<= n;
        have -= n;
        if (have <= 32) {
            buf |= (uint64_t)nextw << (32 - have);
            have += 32;
            nextw = src_word(b, wi);
            wi++;
        }
    }
};

// Decodes one symbol at state s.  Returns false when the code is not in the table (the
// reference would never leave its bit loop, Decoder.cpp:704-748).
// On return: sym, total bits consumed (code + magnitude bits), value bits in `vbits`.
__device__ __forceinline__ bool decode_symbol(const LdsTables& T, uint32_t win, int ti, uint32_t& sym, uint32_t& len)
{
    uint32_t e = T.lut[ti][win >> (32 - LUT_BITS)];
    if (e & LUT_LONG) {  // 10..16-bit code: one more table read
        if (e != LUT_SEARCH) {
            e = T.lut2[ti][e & 0x7FFF][(win >> (32 - 16)) & ((1 << LUT2_BITS) - 1)];
        } else {
            e = 0;
            for (int l = LUT_BITS + 1; l <= 16; ++l) {
                int code = (int)(win >> (32 - l));
                if (code <= T.maxcode[ti][l]) {
                    e = ((uint32_t)l << 8) | T.symbols[ti][(T.valoff[ti][l] + code) & 255];
                    break;
                }
            }
        }
    }
    sym = e & 0xFF;
    len = e >> 8;
    if (e == 0) {
        len = 16;  // no such code: keep moving (only a speculative decode or a corrupt stream gets here)
        return false;
    }
    return true;
}

// JPEG EXTEND (bitStringtoValue, Image.cpp:285-302)
__device__ __forceinline__ int extend(uint32_t bits, uint32_t cat)
{
    if (cat == 0) return 0;
    int v = (int)bits;
    return (bits >> (cat - 1)) ? v : v - (int)((1u << cat) - 1u);
}

struct RunResult {
    uint64_t exit_state;
    int nb;       // blocks started (DC symbols decoded)
    int dc[3];    // sum of DC differences of those blocks
#if KPEG_SYNC_STATS
    uint32_t iters;
#endif
};

// Sync/count run: decode from `s` until the bit position reaches `pend`.
__device__ __forceinline__ RunResult run_count(const LdsTables& T, const BitSrc& w, DecState s, uint32_t pend)
{
    RunResult r;
    r.nb = 0;
    r.dc[0] = r.dc[1] = r.dc[2] = 0;
    BitReader br;
    br.init(w, s.p);
    uint32_t p = s.p, c = s.c, k = s.k;
    int nb = 0, dc0 = 0, dc1 = 0, dc2 = 0;
#if KPEG_SYNC_STATS
    r.iters = 0;
#endif
    // branch-free state machine: lanes of a wavefront sit at different points of their blocks
    while (p < pend) {
        const uint32_t win = br.peek();
        const uint32_t isac = k != 0;
        uint32_t sym, len;
        decode_symbol(T, win, (int)(isac * 2 + (c != 0)), sym, len);
        const uint32_t cat = sym & 15, run = sym >> 4;   // EOB (symbol 0) has cat 0: it consumes len bits
        const uint32_t used = len + cat;
        // DC difference (EXTEND), only counted when this symbol is a DC symbol
        const uint32_t bits = __builtin_amdgcn_ubfe(win << len, 32 - cat, cat);
        const uint32_t half = (1u << cat) >> 1;
        const int d = (int)bits - (bits < half ? (int)((1u << cat) - 1u) : 0);
        const int dd = isac ? 0 : d;
        nb += (int)(isac ^ 1u);
        dc0 += c == 0 ? dd : 0;
        dc1 += c == 1 ? dd : 0;
        dc2 += c == 2 ? dd : 0;
        uint32_t kn = isac ? k + run + 1 : 1u;
        kn = (isac && sym == 0) ? 64u : kn;       // EOB
        const bool done = kn >= 64;                // block complete (EOB, or ACCodesCount == 63, Decoder.cpp:759)
        k = done ? 0u : kn;
        c = done ? (c == 2 ? 0u : c + 1) : c;
        p += used;
        br.consume(w, used);
#if KPEG_SYNC_STATS
        r.iters++;
#endif
    }
    s.p = p;
    s.c = c;
    s.k = k;
    r.nb = nb;
    r.dc[0] = dc0;
    r.dc[1] = dc1;
    r.dc[2] = dc2;
    r.exit_state = pack_state(s);
    return r;
}

__device__ __forceinline__ uint32_t locate_segment(const uint32_t* __restrict__ sub_base, uint32_t nseg, uint32_t i)
{
    // largest r with sub_base[r] <= i
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (sub_base[mid] <= i) lo = mid;
        else hi = mid;
    }
    return lo;
}

struct SubGeom {
    uint32_t seg, li;      // segment and index inside it
    uint32_t pstart, pend; // bit range of the sub-sequence
};
__device__ __forceinline__ SubGeom sub_geom(const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ sub_base,
                                            uint32_t nseg, uint32_t i)
{
    SubGeom g;
    g.seg = nseg > 1 ? locate_segment(sub_base, nseg, i) : 0;
    g.li = i - sub_base[g.seg];
    uint32_t s0 = seg_off[g.seg] * 8, s1 = seg_off[g.seg + 1] * 8;
    g.pstart = s0 + g.li * SUBSEQ_BITS;
    g.pend = min(g.pstart + SUBSEQ_BITS, s1);
    if (g.pstart > s1) g.pstart = s1;
    return g;
}

struct SyncArgs {
    const uint32_t* u;
    const uint32_t* seg_off;
    const uint32_t* sub_base;
    EntropyMeta* meta;
    const EntropyTables* tabs;
    uint64_t* X;       // [nsub_cap] exit state of every sub-sequence (written by its workgroup only)
    uint64_t* Xb;      // [2][nwg_cap] exit state of each workgroup's last sub-sequence, ping-pong by pass
    uint8_t* mv;       // [2][nwg_cap] "my last exit state moved in this pass"
    int4* cnt;         // [nsub_cap] (blocks started, dc sums) of the run that produced X
    int4* wsum;        // [nwg_cap] per-workgroup totals of cnt
    uint32_t nwg_cap;
    int pass;
    uint32_t* status;  // KPEG_SYNC_STATS builds only: words 8..13 collect loop counts
};

__device__ __forceinline__ int4 add4(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Pass 0: every sub-sequence decodes from a guessed boundary (its own first bit, DC of
// component 0), then the workgroup iterates "re-decode from my predecessor's exit state
// where that moved" until nothing moves (Jacobi rounds, LDS only).  The first sub-sequence
// of a restart segment starts from a known state; the first one of a workgroup cannot be
// checked inside the workgroup.
// Pass p >= 1: the first sub-sequence of each workgroup whose predecessor workgroup moved in
// pass p-1 re-decodes from that workgroup's exit state and the change, if any, ripples on.
__global__ __launch_bounds__(SYNC_WG) void k_sync_pass(SyncArgs a)
{
    __shared__ LdsTables T;
    __shared__ uint64_t s_X[SYNC_WG];
    __shared__ uint8_t s_dirty[SYNC_WG + 1];
    __shared__ int4 s_red[SYNC_WG / 64];
    __shared__ uint32_t s_bits[STAGE_LDS];
    const uint32_t nsub = a.meta->nsub;
    const int p = a.pass;
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    const uint32_t i0 = g * SYNC_WG;
    if (i0 >= nsub) return;
    if (p >= 2 && a.meta->moved[p - 1] == 0) return;  // converged
    const uint8_t* mv_prev = a.mv + (size_t)((p & 1) ^ 1) * a.nwg_cap;
    uint8_t* mv_cur = a.mv + (size_t)(p & 1) * a.nwg_cap;
    const uint64_t* Xb_prev = a.Xb + (size_t)((p & 1) ^ 1) * a.nwg_cap;
    uint64_t* Xb_cur = a.Xb + (size_t)(p & 1) * a.nwg_cap;
    if (p >= 1 && (g == 0 || mv_prev[g - 1] == 0)) {
        // nothing upstream moved: my states stand
        if (t == 0) {
            mv_cur[g] = 0;
            Xb_cur[g] = Xb_prev[g];
        }
        return;
    }
    load_tables(&T, a.tabs);
    const uint32_t i = i0 + t;
    const bool valid = i < nsub;
    const uint32_t nseg = a.meta->nseg;
    SubGeom geo;
    geo.seg = 0;
    geo.li = 0;
    geo.pstart = geo.pend = 0;
    if (valid) geo = sub_geom(a.seg_off, a.sub_base, nseg, i);
    // stage this workgroup's slice of the bit string (its sub-sequences are contiguous in u)
    BitSrc src;
    {
        const SubGeom g0 = sub_geom(a.seg_off, a.sub_base, nseg, i0);
        src.g = a.u;
        src.lds = s_bits;
        src.w0 = g0.pstart >> 5;
        const uint32_t total_words = (a.meta->n_u + 3) / 4 + 2;
        src.nw = min((uint32_t)STAGE_TOTAL, total_words > src.w0 ? total_words - src.w0 : 0u);
        stage_bits(s_bits, a.u, src.w0, src.nw);
    }
    __syncthreads();

    uint64_t myX = 0;
    int4 mycnt = make_int4(0, 0, 0, 0);
    bool dirty = false;
    uint64_t bentry = 0;
#if KPEG_SYNC_STATS
    uint32_t st_it = 0;
#endif
    if (p == 0) {
        if (valid) {
            DecState s;
            s.p = geo.pstart;
            s.c = 0;
            s.k = 0;
            RunResult r = run_count(T, src, s, geo.pend);
#if KPEG_SYNC_STATS
            st_it = r.iters;
#endif
            myX = r.exit_state;
            mycnt = make_int4(r.nb, r.dc[0], r.dc[1], r.dc[2]);
            dirty = geo.li != 0 && t > 0;
        }
    } else {
        if (valid) {
            myX = a.X[i];
            mycnt = a.cnt[i];
            dirty = t == 0 && geo.li != 0;
        }
        bentry = Xb_prev[g - 1];
    }
    s_X[t] = myX;
    const uint64_t x_at_entry = myX;

#if KPEG_SYNC_STATS
    uint32_t st_rounds = 0, st_jit = 0;
    {
        uint32_t m = st_it;
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if ((t & 63) == 0) { atomicAdd(&a.status[10], m); atomicAdd(&a.status[11], 1u); }
    }
#endif
    for (;;) {
        if (!__syncthreads_or(dirty)) break;  // also publishes s_X
#if KPEG_SYNC_STATS
        st_rounds++;
#endif
        const uint64_t e = t == 0 ? bentry : s_X[t - 1];
        __syncthreads();
        bool changed = false;
        if (dirty) {
            RunResult r = run_count(T, src, unpack_state(e), geo.pend);
#if KPEG_SYNC_STATS
            st_jit = r.iters;
#endif
            changed = r.exit_state != myX;
            myX = r.exit_state;
            mycnt = make_int4(r.nb, r.dc[0], r.dc[1], r.dc[2]);
            s_X[t] = myX;
        }
#if KPEG_SYNC_STATS
        {
            uint32_t m = dirty ? st_jit : 0;
            for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
            if ((t & 63) == 0) { atomicAdd(&a.status[12], m); atomicAdd(&a.status[13], m ? 1u : 0u); }
        }
#endif
        s_dirty[t + 1] = changed;
        __syncthreads();
        dirty = valid && t > 0 && geo.li != 0 && s_dirty[t];
    }

#if KPEG_SYNC_STATS
    if (t == 0) { atomicAdd(&a.status[8], st_rounds); atomicMax(&a.status[9], st_rounds); }
#endif
    if (valid) {
        a.X[i] = myX;
        a.cnt[i] = mycnt;
    }
    // per-workgroup totals for the scan
    int4 tot = mycnt;
    for (int o = 32; o > 0; o >>= 1) {
        tot.x += __shfl_down(tot.x, o);
        tot.y += __shfl_down(tot.y, o);
        tot.z += __shfl_down(tot.z, o);
        tot.w += __shfl_down(tot.w, o);
    }
    if ((t & 63) == 0) s_red[t >> 6] = tot;
    __syncthreads();
    const uint32_t last = min((uint32_t)SYNC_WG - 1, nsub - 1 - i0);
    if (t == 0) {
        int4 w = s_red[0];
        for (int q = 1; q < SYNC_WG / 64; ++q) w = add4(w, s_red[q]);
        a.wsum[g] = w;
    }
    if (t == last) {
        const bool moved = p == 0 ? true : (myX != x_at_entry);
        Xb_cur[g] = myX;
        mv_cur[g] = moved;
        // the last workgroup has no successor: its movement needs no further pass
        if (moved && i + 1 < nsub) atomicAdd(&a.meta->moved[p], 1u);
    }
}

// ------------------------------------------------------------------------------------------
// scan of (nb, dc0, dc1, dc2): workgroup totals -> exclusive prefix (single workgroup)
__global__ __launch_bounds__(1024) void k_scan_wsum(int4* wsum, EntropyMeta* meta, uint32_t* status, int last_pass)
{
    __shared__ int4 s[1024];
    __shared__ int4 carry;
    const uint32_t nsub = meta->nsub;
    const uint32_t nw = (nsub + SYNC_WG - 1) / SYNC_WG;
    if (threadIdx.x == 0) carry = make_int4(0, 0, 0, 0);
    __syncthreads();
    for (uint32_t base = 0; base < nw; base += 1024) {
        uint32_t i = base + threadIdx.x;
        int4 v = i < nw ? wsum[i] : make_int4(0, 0, 0, 0);
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int4 t = make_int4(0, 0, 0, 0);
            if ((int)threadIdx.x >= o) t = s[threadIdx.x - o];
            __syncthreads();
            s[threadIdx.x] = add4(s[threadIdx.x], t);
            __syncthreads();
        }
        int4 incl = s[threadIdx.x], c = carry;
        if (i < nw) wsum[i] = make_int4(c.x + incl.x - v.x, c.y + incl.y - v.y, c.z + incl.z - v.z, c.w + incl.w - v.w);
        __syncthreads();
        if (threadIdx.x == 1023) carry = add4(c, incl);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        meta->total_blocks = (uint32_t)carry.x;
        uint32_t passes = 1;
        for (int t = 1; t <= last_pass; ++t)
            if (t == 1 || meta->moved[t - 1]) passes = t + 1;
        status[2] = passes;
        if (meta->moved[last_pass] != 0) atomicOr(&status[1], 4u);  // not converged within the enqueued passes
    }
}

// materialised exclusive prefix (only needed when restart segments re-base it)
__global__ __launch_bounds__(SYNC_WG) void k_scan_apply(const int4* cnt, const int4* wsum, const EntropyMeta* meta, int4* prefix)
{
    __shared__ int4 s[SYNC_WG];
    const uint32_t nsub = meta->nsub;
    uint32_t i = blockIdx.x * SYNC_WG + threadIdx.x;
    if (blockIdx.x * SYNC_WG >= nsub) return;
    int4 v = i < nsub ? cnt[i] : make_int4(0, 0, 0, 0);
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < SYNC_WG; o <<= 1) {
        int4 t = make_int4(0, 0, 0, 0);
        if ((int)threadIdx.x >= o) t = s[threadIdx.x - o];
        __syncthreads();
        s[threadIdx.x] = add4(s[threadIdx.x], t);
        __syncthreads();
    }
    int4 incl = s[threadIdx.x], w = wsum[blockIdx.x];
    if (i < nsub) prefix[i] = make_int4(w.x + incl.x - v.x, w.y + incl.y - v.y, w.z + incl.z - v.z, w.w + incl.w - v.w);
}

// ------------------------------------------------------------------------------------------
// K2: write pass
struct WriteArgs {
    const uint32_t* u;
    const uint32_t* seg_off;
    const uint32_t* sub_base;
    const EntropyMeta* meta;
    const EntropyTables* tabs;
    const uint64_t* X;   // [nsub_cap] converged exit states
    const int4* cnt;     // per-sub-sequence (blocks started, dc sums)
    const int4* wsum;    // exclusive prefix of the per-workgroup totals
    const int4* prefix;  // materialised exclusive prefix (restart segments only, else null)
    int16_t* coef;
    float* ebound;       // [block] K4's per-block error bound
    uint32_t nsub_cap;
    uint32_t nmcu;
    uint32_t interval;   // 0 = none
    uint32_t* status;
};

constexpr int WB_STRIDE = 72;  // int16 per lane block buffer: 64 + 8 pad (144 bytes, conflict-free b128)

__global__ __launch_bounds__(256) void k_write(WriteArgs a)
{
    __shared__ LdsTables T;
    __shared__ __attribute__((aligned(16))) int16_t s_blk[256 * WB_STRIDE];
    __shared__ int4 s_pre[256];
    __shared__ uint4 s_flush[4][64];   // per wavefront: (block, error bound, owner lane) of blocks to write
    __shared__ uint32_t s_bits[STAGE_LDS];
    const uint32_t nsub = a.meta->nsub;
    if (blockIdx.x * 256u >= nsub) return;
    load_tables(&T, a.tabs);
    int16_t* blk = s_blk + threadIdx.x * WB_STRIDE;
    {
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4* b4 = reinterpret_cast<uint4*>(blk);
        for (int q = 0; q < 8; ++q) b4[q] = z;
    }
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    BitSrc w;
    {
        const SubGeom g0 = sub_geom(a.seg_off, a.sub_base, a.meta->nseg, blockIdx.x * 256u);
        w.g = a.u;
        w.lds = s_bits;
        w.w0 = g0.pstart >> 5;
        const uint32_t total_words = (a.meta->n_u + 3) / 4 + 2;
        w.nw = min((uint32_t)STAGE_TOTAL, total_words > w.w0 ? total_words - w.w0 : 0u);
        stage_bits(s_bits, a.u, w.w0, w.nw);
    }
    if (!a.prefix) {
        int4 v = i < nsub ? a.cnt[i] : make_int4(0, 0, 0, 0);
        s_pre[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            int4 t = make_int4(0, 0, 0, 0);
            if ((int)threadIdx.x >= o) t = s_pre[threadIdx.x - o];
            __syncthreads();
            s_pre[threadIdx.x] = add4(s_pre[threadIdx.x], t);
            __syncthreads();
        }
        int4 incl = s_pre[threadIdx.x], w = a.wsum[blockIdx.x];
        __syncthreads();
        s_pre[threadIdx.x] = make_int4(w.x + incl.x - v.x, w.y + incl.y - v.y, w.z + incl.z - v.z, w.w + incl.w - v.w);
    }
    __syncthreads();
    // lanes past the last sub-sequence stay: they help to write the others' blocks
    const uint32_t nseg = a.meta->nseg;
    const uint32_t ii = min(i, nsub - 1);
    const SubGeom g = sub_geom(a.seg_off, a.sub_base, nseg, ii);
    const uint64_t* X = a.X;

    DecState s;
    if (g.li == 0) {
        s.p = g.pstart;
        s.c = 0;
        s.k = 0;
    } else {
        s = unpack_state(X[ii - 1]);
    }
    // block index and DC predictors at entry, relative to the segment start
    int4 pre;
    const uint32_t first = a.sub_base[g.seg];
    if (a.prefix) {
        pre = make_int4(0, 0, 0, 0);
        if (g.li != 0) {
            int4 pi = a.prefix[ii], p0 = a.prefix[first];
            pre = make_int4(pi.x - p0.x, pi.y - p0.y, pi.z - p0.z, pi.w - p0.w);
        }
    } else {
        // single segment: exclusive scan of cnt inside the workgroup + the workgroup's offset
        pre = s_pre[threadIdx.x];
    }
    const uint32_t seg_mcu0 = a.interval ? g.seg * a.interval : 0;
    const uint32_t seg_mcus = a.interval ? min(a.interval, a.nmcu - seg_mcu0) : a.nmcu;
    const uint32_t blk_limit = seg_mcus * 3;       // blocks of this segment
    uint32_t b = (uint32_t)pre.x;                  // next block to start, within the segment
    const uint32_t seg_pend = a.seg_off[g.seg + 1] * 8;

    const uint32_t lane = threadIdx.x & 63;
    int16_t* wave_blk = s_blk + (threadIdx.x & ~63u) * WB_STRIDE;
    uint4* fl_q = s_flush[threadIdx.x >> 6];
    BitReader br;
    br.init(w, s.p);
    // One symbol per iteration for every lane (a flat state machine: the lanes of a wavefront sit at
    // different points of different blocks, so nested per-block / per-symbol loops would make every
    // lane wait for the longest block of each round).
    //   own == false : finishing a block that began in an earlier sub-sequence; it belongs to the
    //                  lane that started it, so its symbols are only stepped over
    //   k == 0       : next symbol is a DC symbol (a block starts, if this sub-sequence still has bits)
    uint32_t err = 0;
    uint32_t p = s.p, k = s.k;
    bool own = k == 0;
    uint32_t cb = b % 3;                 // component of block b (== s.c at a block start of a valid stream)
    uint32_t c = own ? cb : s.c;         // component whose tables are in use
    int pred0 = pre.y, pred1 = pre.z, pred2 = pre.w;
    bool keep_ac = false;
    uint64_t touched = 0;
    float Asum = 0.0f;   // K4's error bound for this block: A = sum |in|, nnz = non-zero AC terms (idct_colour.hip.h)
    int nnz = 0;
    bool active = i < nsub;
    for (;;) {
        const bool indc = k == 0;
        active = active && (indc ? (p < g.pend && b < blk_limit) : (own || p < seg_pend));
        if (!__any(active)) break;   // wave-uniform: every lane stays for the cooperative write-out
        const uint32_t tdc = c ? 1u : 0u;
        bool done = false;
        if (active) {
        const uint32_t win = br.peek();
        uint32_t sym, len;
        const bool ok = decode_symbol(T, win, (int)((indc ? 0u : 2u) + tdc), sym, len);
        if (!ok && own) err |= 8;
        const uint32_t cat = sym & 15, run = sym >> 4;
        const uint32_t bits = __builtin_amdgcn_ubfe(win << len, 32 - cat, cat);
        const uint32_t half = (1u << cat) >> 1;
        int val = (int)bits - (bits < half ? (int)((1u << cat) - 1u) : 0);   // EXTEND; 0 when cat == 0
        const uint32_t used = len + cat;   // EOB and "no such code" have cat 0
        p += used;
        br.consume(w, used);
        uint32_t nat = 0;
        bool place;
        if (indc) {
            if (run) err |= 16;  // DC symbol with a run nibble: outside the contract
            // DCDiff[c] += zz[0]   (MCU.cpp:107)
            pred0 += c == 0 ? val : 0;
            pred1 += c == 1 ? val : 0;
            pred2 += c == 2 ? val : 0;
            val = c == 0 ? pred0 : (c == 1 ? pred1 : pred2);
            keep_ac = sym != 0;  // quirk Q1: a DC "EOB" drops the block's AC terms (MCU.cpp:97-100)
            own = true;
            touched = 0;
            nnz = 0;
            Asum = fabsf(0.25f * (0x1.fffffep-2f * ((float)val * T.q00[tdc])));
            k = 1;               // coefficients placed so far + 1
            place = true;
            done = false;
        } else {
            const uint32_t kn = k + run + 1;
            const bool eob = sym == 0;
            const bool over = kn > 64;
            if (over && !eob && own) err |= 32;  // run past the end of the block
            place = !eob && !over && own && keep_ac;
            nat = T.zz[(kn - 1) & 63];
            done = eob || kn >= 64;
            k = kn;
            if (place) {
                touched |= 1ull << nat;
                Asum += fabsf((float)val * T.mscale[tdc][nat]);
                nnz += val != 0;
            }
        }
#if KPEG_ABLATE_W != 3
        if (place) blk[nat] = (int16_t)val;
#endif
        }
        // Blocks completed in this iteration leave together: the owners queue (block, bound) and the
        // wavefront writes the queued 128-byte blocks eight lanes to a block, so every store
        // instruction fills whole cache lines (a lane flushing its own block alone would send eight
        // 16-byte partial-line writes to L2: measured 2x the kernel's whole decode time).
        const bool fl = done && own;   // done is false on lanes that sat this iteration out
        const uint64_t flmask = __ballot(fl);
        if (fl) {
            if (p > seg_pend + 32) err |= 64;  // ran off the end of the data
            const uint32_t gb = seg_mcu0 * 3 + b;
            // == block_ebound() of idct_colour.hip.h (range guard: +inf sends the whole block to the exact path)
            const float E = nnz ? (0x1.004p-24f * Asum) * ((float)nnz + 14.5f) : 0.0f;
            // sign bit: all non-zero AC terms in the 2x2 corner (natural positions 1, 8, 9)
            const bool corner = (touched & ~0x302ull) == 0;
            const float eb = !(Asum < (tdc ? 249.0f : 31000.0f)) ? __builtin_inff() : (corner ? -E : E);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(flmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)flmask, 0));
            fl_q[rank] = make_uint4(gb, __float_as_uint(eb), lane, 0);
            b++;
            cb = cb == 2 ? 0u : cb + 1;
        }
        if (flmask) {   // wave-uniform
            const uint32_t nfl = (uint32_t)__popcll(flmask);
            for (uint32_t j = lane >> 3; j < nfl; j += 8) {
                const uint4 q = fl_q[j];
                uint4* src = reinterpret_cast<uint4*>(wave_blk + q.z * WB_STRIDE) + (lane & 7);
#if KPEG_ABLATE_W != 1 && KPEG_ABLATE_W != 2
#if KPEG_ABLATE_W == 4
                reinterpret_cast<uint4*>(a.coef + (size_t)(q.x & 4095) * 64)[lane & 7] = *src;
                if ((lane & 7) == 0) a.ebound[q.x & 4095] = __uint_as_float(q.y);
#elif KPEG_ABLATE_W == 5
                if ((lane & 7) == 0) a.ebound[q.x] = __uint_as_float(q.y);
#elif KPEG_ABLATE_W == 7
                reinterpret_cast<uint4*>(a.coef + (size_t)q.x * 64)[lane & 7] = make_uint4(q.x, q.y, q.z, 1);
#elif KPEG_ABLATE_W == 8
                if ((lane & 7) == 0) a.ebound[q.x] = __uint_as_float(q.y + src->x);
#elif KPEG_ABLATE_W == 6
                reinterpret_cast<uint4*>(a.coef + (size_t)q.x * 64)[lane & 7] = *src;
#else
                reinterpret_cast<uint4*>(a.coef + (size_t)q.x * 64)[lane & 7] = *src;
                if ((lane & 7) == 0) a.ebound[q.x] = __uint_as_float(q.y);
#endif
#endif
#if KPEG_ABLATE_W != 2
                *src = make_uint4(0, 0, 0, 0);
#endif
            }
        }
        if (done) {
            k = 0;
            c = cb;
            own = true;
        }
    }
    // the last sub-sequence of a segment must have produced the segment's last block
    if (i < nsub && g.li + 1 == a.sub_base[g.seg + 1] - first && b < blk_limit) err |= 128;
    if (err) atomicOr(&a.status[1], err);
}

// ------------------------------------------------------------------------------------------
static int ent_grow(void** p, size_t* cap, size_t need, hipStream_t stream, std::string* err)
{
    if (need <= *cap) return KPEG_HIP_OK;
    if (*p) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(*p);
        *p = nullptr;
        *cap = 0;
    }
    size_t want = need + need / 4 + 4096;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        *err = std::string("hipMalloc(entropy scratch): ") + hipGetErrorString(e);
        return KPEG_HIP_E_NOMEM;
    }
    *cap = want;
    return KPEG_HIP_OK;
}

// ev: kpeg_hip_ctx::ev (EV_UNSTUFF=1, EV_SYNC=2, EV_SCAN=3, EV_WRITE=4, EV_DC=5) or null
static int entropy_decode_launch(EntropyScratch* S, const EntropyTables& tabs, const EntropyLaunch& L, hipEvent_t* ev, bool* ev_rec,
                                 std::string* err)
{
#define ENT_HIP(expr)                                                      \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) {                                            \
            *err = std::string(#expr) + ": " + hipGetErrorString(_e);      \
            return KPEG_HIP_E_DEVICE;                                      \
        }                                                                  \
    } while (0)
    auto mark = [&](int which) {
        if (ev && hipEventRecord(ev[which], L.stream) == hipSuccess) ev_rec[which] = true;
    };
    if (L.scan_len >= (1ull << 28)) {
        *err = "entropy-coded segment larger than 256 MiB";
        return KPEG_HIP_E_UNSUPPORTED;
    }
    const uint32_t n = (uint32_t)L.scan_len;
    const uint32_t nseg_expected = L.restart_interval ? (L.nmcu + L.restart_interval - 1) / L.restart_interval : 1;
    const uint32_t nparts = (n + US_BLOCK_BYTES - 1) / US_BLOCK_BYTES;
    const uint32_t nsub_cap = (uint32_t)(((uint64_t)n * 8 + SUBSEQ_BITS - 1) / SUBSEQ_BITS) + nseg_expected + 1;
    const uint32_t seg_cap = nseg_expected + 2;
    int rc;
    if ((rc = ent_grow(&S->d_u, &S->u_cap, (size_t)n + 64, L.stream, err))) return rc;
    if ((rc = ent_grow(&S->d_part, &S->part_cap, (size_t)nparts * sizeof(uint2), L.stream, err))) return rc;
    if ((rc = ent_grow(&S->d_segoff, &S->seg_cap, (size_t)seg_cap * 2 * sizeof(uint32_t), L.stream, err))) return rc;
    if ((rc = ent_grow(&S->d_state, &S->state_cap, (size_t)nsub_cap * 8 + ((size_t)nsub_cap / SYNC_WG + 2) * 18 + 64, L.stream, err))) return rc;
    if ((rc = ent_grow(&S->d_cnt, &S->cnt_cap, (size_t)nsub_cap * 32, L.stream, err))) return rc;
    if ((rc = ent_grow(&S->d_wsum, &S->wsum_cap, ((size_t)nsub_cap / SYNC_WG + 2) * 16, L.stream, err))) return rc;
    if (!S->d_meta) ENT_HIP(hipMalloc((void**)&S->d_meta, sizeof(EntropyMeta)));
    if (!S->d_tabs) ENT_HIP(hipMalloc((void**)&S->d_tabs, sizeof(EntropyTables)));
    if (!S->tabs_valid || std::memcmp(&S->h_tabs_cached, &tabs, sizeof(tabs)) != 0) {
        // tables changed: earlier launches may still read the old ones
        ENT_HIP(hipStreamSynchronize(L.stream));
        S->h_tabs_cached = tabs;
        ENT_HIP(hipMemcpy(S->d_tabs, &S->h_tabs_cached, sizeof(tabs), hipMemcpyHostToDevice));
        S->tabs_valid = true;
    }

    uint32_t* seg_off = (uint32_t*)S->d_segoff;
    uint32_t* sub_base = seg_off + seg_cap;
    const uint32_t nwg_cap = (nsub_cap + SYNC_WG - 1) / SYNC_WG;
    uint64_t* X = (uint64_t*)S->d_state;
    uint64_t* Xb = X + nsub_cap;
    uint8_t* mv = (uint8_t*)(Xb + 2 * (size_t)nwg_cap);
    int4* cnt = (int4*)S->d_cnt;
    int4* prefix = cnt + nsub_cap;
    const int rst = L.restart_interval ? 1 : 0;

    hipLaunchKernelGGL(k_unstuff_count, dim3(nparts), dim3(US_THREADS), 0, L.stream, L.d_scan, n, rst, (uint2*)S->d_part);
    hipLaunchKernelGGL(k_unstuff_scan, dim3(1), dim3(1024), 0, L.stream, (uint2*)S->d_part, nparts, S->d_meta);
    hipLaunchKernelGGL(k_unstuff_scatter, dim3(nparts), dim3(US_THREADS), 0, L.stream, L.d_scan, n, rst, (const uint2*)S->d_part,
                       (uint8_t*)S->d_u, seg_off, seg_cap);
    hipLaunchKernelGGL(k_seg_setup, dim3(1), dim3(1024), 0, L.stream, S->d_meta, seg_off, sub_base, seg_cap, nsub_cap,
                       (uint8_t*)S->d_u, nseg_expected, L.d_status);
    mark(1);

    SyncArgs sa;
    sa.u = (const uint32_t*)S->d_u;
    sa.seg_off = seg_off;
    sa.sub_base = sub_base;
    sa.meta = S->d_meta;
    sa.tabs = S->d_tabs;
    sa.X = X;
    sa.Xb = Xb;
    sa.mv = mv;
    sa.cnt = cnt;
    sa.wsum = (int4*)S->d_wsum;
    sa.nwg_cap = nwg_cap;
    sa.status = L.d_status;
    const int npass = L.sync_passes > 0 ? L.sync_passes : SYNC_PASSES;
    for (int t = 0; t < npass; ++t) {
        sa.pass = t;
        hipLaunchKernelGGL(k_sync_pass, dim3(nwg_cap), dim3(SYNC_WG), 0, L.stream, sa);
    }
    mark(2);

    hipLaunchKernelGGL(k_scan_wsum, dim3(1), dim3(1024), 0, L.stream, (int4*)S->d_wsum, S->d_meta, L.d_status, npass - 1);
    if (rst)
        hipLaunchKernelGGL(k_scan_apply, dim3(nwg_cap), dim3(SYNC_WG), 0, L.stream, (const int4*)cnt, (const int4*)S->d_wsum,
                           (const EntropyMeta*)S->d_meta, prefix);
    mark(3);

    WriteArgs wa;
    wa.u = (const uint32_t*)S->d_u;
    wa.seg_off = seg_off;
    wa.sub_base = sub_base;
    wa.meta = S->d_meta;
    wa.tabs = S->d_tabs;
    wa.X = X;
    wa.cnt = cnt;
    wa.wsum = (const int4*)S->d_wsum;
    wa.prefix = rst ? prefix : nullptr;
    wa.coef = L.d_coef;
    wa.ebound = L.d_ebound;
    wa.nsub_cap = nsub_cap;
    wa.nmcu = L.nmcu;
    wa.interval = L.restart_interval;
    wa.status = L.d_status;
    hipLaunchKernelGGL(k_write, dim3(nwg_cap), dim3(256), 0, L.stream, wa);
    mark(4);
    mark(5);
    ENT_HIP(hipGetLastError());
    return KPEG_HIP_OK;
#undef ENT_HIP
}

}  // namespace kpeg_dev
