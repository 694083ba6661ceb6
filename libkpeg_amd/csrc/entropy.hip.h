// libkpeg_amd/csrc/entropy.hip.h -- K0..K2: on-device entropy decode for gfx950.
//
// Replaces JPEGDecoder::byteStuffScanData (src/Decoder.cpp:621-653), the Huffman bit loop
// of JPEGDecoder::decodeScanData (:694-803) with HuffmanTree::contains
// (src/HuffmanTree.cpp:164-193) and bitStringtoValue (src/Image.cpp:285-302), and the RLE
// walk + DC prediction of MCU::constructMCU (src/MCU.cpp:91-108) of the reference.
//
// A baseline scan without restart markers is one serial bit string, so the decode is the
// two-pass "self-synchronising sub-sequence" scheme:
//   (stage)       one image without restart markers needs no kernel for it: the sub-sequences are chunks of
//                 SUBSEQ_BITS / 8 bytes of the scan as it is, and the workgroups of K1 and K2 remove the 00 after FF
//                 (reference rule incl. its tail rule) from the chunks they stage, while they stage them
//                 (stage_unstuff; positions that leave a workgroup are chunk << 7 | bit).
//   K0  unstuff   restart segments and batches: byte-parallel removal of the 00 after FF and, with DRI, of the
//                 RSTn markers; compaction by decoupled look-back, one launch.  Output is the bit string as
//                 big-endian 32-bit words + restart-segment offsets (the images of a fused batch are such segments).
//   K1  sync      one item per sub-sequence, one per thread: decode from a guessed codeword boundary, then every
//                 wavefront settles its 64 consecutive items on its own -- an item decodes again exactly when its left
//                 neighbour's exit state (one lane over: a DPP shift) differs from the state it last decoded from
//                 (Huffman streams re-synchronise after a few symbols), no barrier, no work list.  These decodes want
//                 the exit state only (run_exit): the kernel lasts as long as its longest chain of them, one lane after
//                 the other, so their symbol step is as short as it can be made and takes AC symbols two at a time where
//                 the table has them (EntropyTables::lutx).  When the entry states stand, every item decodes once more
//                 and counts: the blocks it starts, their DC differences and (compact stream) the records K2 will write.
//                 A workgroup also decodes the last WARM sub-sequences of its predecessor, which gives it its own entry
//                 state without waiting for the predecessor; that assumption is checked against the predecessor's real
//                 exit state afterwards.  The first sub-sequence of every restart segment starts from a known state.
//                 Pass 0 clears the dense coefficient buffer in the background.
//                 A second launch verifies them, all workgroups in parallel, and only a workgroup that guessed wrong
//                 decodes again (synthetic fields never have one; photographs do, a few per cent of their workgroups).
//                 The third and last launch is chained (every workgroup waits for its predecessor's published state) if
//                 the second still moved something -- no stream is given up -- and in any case runs the
//       scan      exclusive prefix sum of (blocks, dc[3], records) over the workgroups: with K2's local
//                 scan, absolute block index, DC predictors and record ordinal at every sub-sequence entry.
//   K1 + K2 in one kernel (k_sync_write) where it applies -- one image without restart markers on the compact stream, up
//                 to 4096 workgroups, either sub-sequence size: a workgroup does K1's pass 0, publishes, checks the entry state
//                 it assumed against its predecessor's exit state (and does K1's work again from the right one if it was
//                 wrong, publishing its totals a second time), adds up its predecessors' totals and writes (K2's core) from
//                 the bits and tables it has in LDS.  The kernel is enqueued twice: its second, strict launch leaves at once
//                 unless the first gave the call up (a stream that never re-synchronises inside a workgroup, an expired wait).
//   K2  write     one lane per sub-sequence decodes its own symbols again from its true entry
//                 state and writes what K4 reads -- the non-zero coefficients scattered into the cleared dense
//                 buffer (natural order, absolute DC, quirk Q1 applied), or, compact stream, a 4-byte record per
//                 non-zero AC coefficient at the lane's next ordinal (four at a time through a ring in LDS) + the
//                 blocks' DC values + every tile's first record -- plus K4's
//                 per-block error bound (blocks split over lanes or workgroups are settled by
//                 the side that holds their end).
// The decode loops are bound by the instruction count of one symbol step (a wavefront runs
// ~5 cycles per instruction on its own), so a Huffman table entry is a ready-made 32-bit
// record (bits used, coefficient advance, flags) and the decoder's state is the LDS address
// of the table in use.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <string>

#include "../../include/kpeg_hip.h"
#include "kpeg_tables.h"

namespace kpeg_dev {

#ifndef KPEG_SYNC_STATS
#define KPEG_SYNC_STATS 0   // 1: per-wavefront timelines of K1's pass 0 and of K2 into g_ent_stamp (tools/sync_dbg.py)
#endif
#ifndef KPEG_SUBSEQ_BITS
#define KPEG_SUBSEQ_BITS 96
#endif
#ifndef KPEG_SYNC_WG
#define KPEG_SYNC_WG 512
#endif
#ifndef KPEG_SUBSEQ_BITS_DENSE
#define KPEG_SUBSEQ_BITS_DENSE 384
#endif
// Bits per sub-sequence: a multiple of 32, >= 64.  K1/K2 exist for two sizes and the host picks per call from
// the stream's bit rate: 96 (best on the 8K q75 workload, ~1 bit per pixel) and 384 for dense streams (from
// 3 bits per pixel: they re-synchronise over thousands of bits, and fewer, longer rounds halve K1's time).
constexpr int SUBSEQ_SPARSE = KPEG_SUBSEQ_BITS, SUBSEQ_DENSE = KPEG_SUBSEQ_BITS_DENSE;
// A third size since the end of round 3, for SMALL sparse pictures: a picture of a few megapixels is a handful of workgroups on a
// chip of 256 CUs, and what it costs is the length of one lane's chain of symbols -- decoded three times over (K1's exit states,
// K1's counts, K2) at ~0.85 us a symbol when nothing else hides the latencies.  64-bit sub-sequences are two thirds of that chain:
// 512x512 ... 2560x1440 at 1 bit/px 9-12 % faster end to end, photographs of 640x424 at 1-1.4 bit/px 13-14 %; from ~3 bits per pixel
// the extra rounds cost more than the shorter chain saves (tools/small_images_layout.py, profiles/r03_l_small_pictures_layout.txt).
#ifndef KPEG_SUBSEQ_BITS_SMALL
#define KPEG_SUBSEQ_BITS_SMALL 64
#endif
constexpr int SUBSEQ_SMALL = KPEG_SUBSEQ_BITS_SMALL < KPEG_SUBSEQ_BITS ? KPEG_SUBSEQ_BITS_SMALL : KPEG_SUBSEQ_BITS;   // (== SUBSEQ_SPARSE: no third size)
constexpr int SYNC_WG = KPEG_SYNC_WG;          // threads per workgroup of K1 and K2
constexpr int SYNC_PASSES = 3;   // sync kernels enqueued per call: pass 0, the verifying pass 1, the chained pass (+ scan of the totals)
#ifndef KPEG_WARM_BITS
#define KPEG_WARM_BITS 1024
#endif
constexpr int WARM_BITS = KPEG_WARM_BITS;  // a workgroup decodes this much of its predecessor's tail to find its own entry state
// The dense sub-sequences have their own figure (KPEG_WARM_BITS_DENSE) because it was measured separately, and is the same 1024 bits
// = three items: round 3 had six for a while (separate launches: an 8K photograph at 3.5 bit/px 0.278 -> 0.244 ms in K1, a 4K field
// at 11 bit/px 0.227 -> 0.184) until those streams took k_sync_write, whose workgroups repair a wrong entry state themselves --
// photographs at 2.9-3.8 bit/px then measure the same with three or six (+-5 % either way), the 4K field loses 10 % and the q95-noise
// stress figure gains 19 % (1.68 -> 1.36 ms: not the lead-in's length but 509 own items per workgroup instead of 506; not understood).
#ifndef KPEG_WARM_BITS_DENSE
#define KPEG_WARM_BITS_DENSE KPEG_WARM_BITS
#endif
constexpr int WARM_BITS_DENSE = KPEG_WARM_BITS_DENSE;
constexpr int ITEMS = SYNC_WG;
// the geometry that follows from the sub-sequence size S, as local constants of the code templated on S
#define KPEG_GEOMETRY(S)                                                                                                  \
    constexpr int SUBSEQ_BITS = (S), SUBSEQ_WORDS = (S) / 32;                                                             \
    constexpr int WARM = (((S) >= SUBSEQ_DENSE && SUBSEQ_DENSE > SUBSEQ_SPARSE ? WARM_BITS_DENSE : WARM_BITS) + (S) - 1) / (S); /* warm-up sub-sequences */ \
    constexpr int OWN = SYNC_WG - WARM; /* sub-sequences per workgroup: with the warm-up ones a thread each */            \
    static_assert(OWN >= WARM && OWN >= 64, "workgroup too small for the warm-up distance");                              \
    static_assert((S) >= 64 && (S) % 32 == 0, "a symbol (<= 31 bits) must not jump over a whole sub-sequence");          \
    (void)SUBSEQ_BITS;                                                                                                    \
    (void)SUBSEQ_WORDS

#ifndef KPEG_LUT_BITS
#define KPEG_LUT_BITS 9
#endif
constexpr int LUT_BITS = KPEG_LUT_BITS;
constexpr int LUT2_BITS = 16 - LUT_BITS;  // remaining bits of a long code
#ifndef KPEG_POOL_SUBS
#define KPEG_POOL_SUBS 24
#endif
constexpr int POOL_SUBS = KPEG_POOL_SUBS;             // second-level tables shared by the four Huffman tables

// Table entry: everything one symbol step needs, ready-made.
//   [4:0] bits used (code + magnitude)   [9:5] code length   [13:10] magnitude bits (category)
//   [14] entry of a DC table             [15] symbol 0x00 (DC: quirk Q1, AC: EOB) or no such code
//   [22:16] coefficient advance: AC run + 1, EOB / no such code 64, DC 65 (>= 64 ends the table's turn)
//   [23] no such code                    [24] DC symbol with a run nibble (outside the contract)
//   [25] DC symbol other than 0x00: the block keeps its AC terms (quirk Q1)
//   [26] AC symbol that carries a non-zero coefficient (category > 0): one record of the compact coefficient stream
//   [27] AC symbol other than EOB (a run: with a coefficient, or ZRL): must not run past the end of its block
//   [30:28] coefficient index after the table's turn: 1 behind a DC symbol, 0 behind a block (run_exit reads it; its
//        two-symbol entries, EntropyTables::lutx, also have 2..7 there, and [24] set = the turn of TWO tables ends)
//   [31] code longer than LUT_BITS: [15:0] = second-level table in the pool, E_SEARCH = none left
constexpr uint32_t E_ISDC = 1u << 14, E_ZERO = 1u << 15, E_BAD = 1u << 23, E_DCRUN = 1u << 24, E_KEEP = 1u << 25, E_REC = 1u << 26, E_ACSYM = 1u << 27,
                   E_LONG = 1u << 31;
constexpr uint32_t E_SEARCH = 0xFFFFu;

__host__ __device__ inline uint32_t make_entry(uint32_t len, uint32_t sym, bool isdc)
{
    const uint32_t cat = sym & 15, run = sym >> 4;
    const uint32_t kadv = isdc ? 65u : (sym == 0 ? 64u : run + 1);
    return (len + cat) | (len << 5) | (cat << 10) | (isdc ? E_ISDC : 0u) | (sym == 0 ? E_ZERO : 0u) | (kadv << 16) |
           ((isdc && run) ? E_DCRUN : 0u) | ((isdc && sym != 0) ? E_KEEP : 0u) | ((!isdc && cat != 0) ? E_REC : 0u) |
           ((!isdc && sym != 0) ? E_ACSYM : 0u) | (isdc ? 1u << 28 : 0u);
}
// no such code: keep moving by 16 bits (only a speculative decode or a corrupt stream gets here;
// the reference would never leave its bit loop, Decoder.cpp:704-748)
__host__ __device__ inline uint32_t bad_entry(bool isdc)
{
    return 16u | (16u << 5) | (isdc ? E_ISDC : 0u) | E_ZERO | ((isdc ? 65u : 64u) << 16) | E_BAD | (isdc ? 1u << 28 : 0u);
}

struct EntropyTables {  // built on the host per frame, copied to the device when it changes
    uint32_t lut[4][1 << LUT_BITS];          // [class*2+id], indexed by the next LUT_BITS bits
    uint32_t pool[POOL_SUBS][1 << LUT2_BITS];  // second level: the LUT2_BITS bits after a long code's prefix
    int32_t maxcode[4][18];          // canonical: largest code of each length (-1 if none)
    int32_t valoff[4][18];           // symbol index of the first code of each length minus that code
    uint8_t symbols[4][256];
    uint8_t zz[64];                  // zig-zag -> natural
    float mscale_zz[2][64];          // 0.25 * cc[u][v] * Q[u][v] by zig-zag position: K4's input scale
    float2 zzm[2][64];               // K2's one read per coefficient: .x = mscale_zz, .y (bits) = natural position << 8 | outside the 2x2 corner << 31
    float q00[2];                    // Q[0][0] of both tables
    float pad16[2];                  // (the tables are copied to LDS in 16-byte pieces)
    // K1's exit-state decodes only (run_exit): the four tables again, [class * 2 + id], with TWO symbols per entry wherever
    // the second symbol's code still lies inside the LUT_BITS window.  Same fields as a one-symbol entry, read the same way:
    // bits used = both symbols'; AC AC: coefficient advance = the sum (an EOB's 64 included), only looked up while k < 48,
    // where the first symbol (advance <= 16) cannot end the block; DC AC: advance 65 (the DC table's turn ends), index
    // after it 1 + the AC symbol's advance (<= 7), Q1 flag of the DC symbol; DC EOB: the same with index 0, no flag, and
    // bit 24: the turn of two tables ends.  (Bit 24 is E_DCRUN in a one-symbol DC entry: never set in these tables.)
    uint32_t lutx[4][1 << LUT_BITS];
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
            t->mscale_zz[tq][k] = 0.25f * cc * (float)f->qt[tq][k];
        }
        t->mscale_zz[tq][0] = 0.25f * (0x1.fffffep-2f * (float)f->qt[tq][0]);  // the reference's own cc[0][0] (Transform.cpp), as K4 uses it
        t->q00[tq] = (float)f->qt[tq][0];
        for (int k = 0; k < 64; ++k) {
            const uint32_t bits = ((uint32_t)KPEG_ZZ_TO_NATURAL[k] << 8) | ((k == 0 || k == 1 || k == 2 || k == 4) ? 0u : 0x80000000u);
            t->zzm[tq][k].x = t->mscale_zz[tq][k];
            std::memcpy(&t->zzm[tq][k].y, &bits, 4);
        }
    }
    int nsub = 0;
    for (int cls = 0; cls < 2; ++cls)
        for (int id = 0; id < 2; ++id) {
            const kpeg_dht& h = f->dht[cls][id];
            const int ti = cls * 2 + id;
            const bool isdc = cls == 0;
            for (int j = 0; j < (1 << LUT_BITS); ++j) t->lut[ti][j] = bad_entry(isdc);
            int code = 0, k = 0;
            for (int len = 1; len <= 16; ++len) {
                int cnt = h.counts[len - 1];
                if (k + cnt > 256) return -1;
                t->valoff[ti][len] = k - code;
                t->maxcode[ti][len] = cnt ? code + cnt - 1 : -1;
                if (cnt && code + cnt - 1 >= (1 << len)) return -1;  // not a prefix code
                for (int i = 0; i < cnt; ++i) {
                    uint8_t sym = h.symbols[k];
                    t->symbols[ti][k] = sym;
                    const uint32_t e = make_entry((uint32_t)len, sym, isdc);
                    if (len <= LUT_BITS) {
                        int first = code << (LUT_BITS - len);
                        for (int j = 0; j < (1 << (LUT_BITS - len)); ++j) t->lut[ti][first + j] = e;
                    } else {
                        const int prefix = code >> (len - LUT_BITS);
                        uint32_t& l1 = t->lut[ti][prefix];
                        if (!(l1 & E_LONG)) {
                            if (nsub < POOL_SUBS) {
                                for (int j = 0; j < (1 << LUT2_BITS); ++j) t->pool[nsub][j] = bad_entry(isdc);
                                l1 = E_LONG | (uint32_t)nsub++;
                            } else {
                                l1 = E_LONG | E_SEARCH;
                            }
                        }
                        if ((l1 & 0xFFFFu) != E_SEARCH) {
                            const int sub = (int)(l1 & 0xFFFFu), rem = len - LUT_BITS;
                            const int first = (code & ((1 << rem) - 1)) << (LUT2_BITS - rem);
                            for (int j = 0; j < (1 << (LUT2_BITS - rem)); ++j) t->pool[sub][first + j] = e;
                        }
                    }
                    code++;
                    k++;
                }
                code <<= 1;
            }
            if (k == 0) return -1;
        }
    for (int id = 0; id < 2; ++id)
        for (int j = 0; j < (1 << LUT_BITS); ++j) {
            // the symbol behind the first one, if its code lies inside the window: always an entry of the AC table
            auto second = [&](uint32_t len1, uint32_t* e2) -> bool {
                if (len1 >= (uint32_t)LUT_BITS) return false;
                *e2 = t->lut[2 + id][(j << len1) & ((1 << LUT_BITS) - 1)];
                return !(*e2 & (E_LONG | E_BAD)) && ((*e2 >> 5) & 31) <= LUT_BITS - len1;
            };
            uint32_t e2 = 0;
            const uint32_t a1 = t->lut[2 + id][j];
            uint32_t x = a1;
            if (!(a1 & (E_LONG | E_BAD | E_ZERO)) && (a1 & E_ACSYM) && second(a1 & 31, &e2))
                x = ((a1 & 31) + (e2 & 31)) | ((((a1 >> 16) & 127) + ((e2 >> 16) & 127)) << 16);
            t->lutx[2 + id][j] = x;
            const uint32_t d1 = t->lut[id][j];
            x = d1 & ~E_DCRUN;
            if (!(d1 & (E_LONG | E_BAD | E_DCRUN)) && second(d1 & 31, &e2)) {
                const uint32_t adv2 = (e2 >> 16) & 127, len = (d1 & 31) + (e2 & 31);
                if (e2 & E_ZERO) x = len | (65u << 16) | E_ISDC | E_DCRUN;                                  // DC EOB
                else if (adv2 <= 6) x = len | (65u << 16) | E_ISDC | (d1 & E_KEEP) | ((1u + adv2) << 28);   // DC AC
            }
            t->lutx[id][j] = x;
        }
    return 0;
}

// Cross-lane moves inside the VALU (DPP) instead of through the LDS crossbar (ds_bpermute, which is what __shfl_* become):
// a scan step is then one add, not a round trip of a hundred cycles -- it matters where one wavefront works alone.
// All 64 lanes must be active.  Lanes a DPP control does not write read 0 (old = 0, bound_ctrl off).
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int dpp0(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, false);
}
// inclusive prefix sum over the wavefront (lane 63 ends with the total)
__device__ __forceinline__ int wave_scan_incl(int x)
{
    int s = x;
    s += dpp0<0x111>(x);               // row_shr:1
    s += dpp0<0x112>(x);               // row_shr:2
    s += dpp0<0x113>(x);               // row_shr:3: four terms, inside every row of 16
    s += dpp0<0x114, 0xF, 0xE>(s);     // row_shr:4 into lanes 4..15 of the rows
    s += dpp0<0x118, 0xF, 0xC>(s);     // row_shr:8 into lanes 8..15
    s += dpp0<0x142, 0xA, 0xF>(s);     // row_bcast:15: rows 1 and 3 take the total of the row before
    s += dpp0<0x143, 0xC, 0xF>(s);     // row_bcast:31: rows 2 and 3 take the total of rows 0..1
    return s;
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) { return (uint32_t)wave_scan_incl((int)x); }
// the value of the lane before (lane 0: 0)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x) { return (uint32_t)dpp0<0x138>((int)x); }   // wave_shr:1

struct EntropyMeta {   // device-resident bookkeeping written by K0
    uint32_t n_u;      // un-stuffed length in bytes
    uint32_t nseg;     // restart segments found (markers + 1)
    uint32_t nsub;     // total sub-sequences
    uint32_t moved[SYNC_PASSES + 8];  // per pass >= 1: workgroups whose last exit state moved
    uint32_t total_blocks;
    uint32_t total_rec;   // records of the compact coefficient stream (K1's count)
    uint32_t ticket;   // workgroups that have finished the chained pass
    uint32_t k0_slot[64], k0_top;   // K0's two-level ticket (restart segments: the last workgroup to finish sets the segments up)
    uint32_t k1_order;   // K1's chained pass: logical workgroup index = the order in which workgroups start (K0 clears it)
    uint32_t fused_fail; // == the call's number: k_sync_write could not finish the call (the three launches behind it do)
    uint32_t repaired;   // workgroups of k_sync_write that found their entry state wrong and put it right themselves (ever: tools read the difference)
    uint32_t strict_order;   // k_sync_write's second launch: logical workgroup index = the order in which its workgroups start (the first launch clears it)
};

// Waits between workgroups never rest on the order in which the hardware dispatches blockIdx (HIP promises none):
//  * K0's decoupled look-back waits a short while for a predecessor's published aggregate and then computes that
//    aggregate ITSELF from the predecessor's input bytes (look-back with fallback): no wait can last, whatever runs
//    or does not run beside this workgroup;
//  * K1's chained pass (a predecessor's result depends on the whole chain before it: no fallback possible) takes its
//    logical index from an atomic ticket when the workgroup starts, so every lower index belongs to a workgroup that
//    is running or done, and bounds the wait in time (s_memrealtime ticks, 100 MHz): on expiry it sets
//    KPEG_ERR_TIMEOUT in the status word and goes on with what it has, so that kpeg_hip_sync() returns
//    KPEG_HIP_E_DEVICE instead of the queue hanging.
constexpr unsigned long long K0_SPIN_TICKS = 50000ull;        // 0.5 ms (a predecessor's 8 KiB take microseconds), then the fallback
constexpr unsigned long long K1_SPIN_TICKS = 2000000000ull;   // 20 s: a predecessor's wait includes the whole chain before it
constexpr uint32_t FUSED_MAX_WG = 4096;   // k_sync_write: every workgroup reads every record before its own -- 8 M records at this size, a few per cent of such a call's time
constexpr unsigned long long FUSED_SPIN_TICKS = 30000ull, FUSED_SPIN_TICKS_DENSE = 100000ull;     // 0.3 ms (three times what the kernel takes on an 8K image; 1 ms with the long sub-sequences, whose workgroups take four times as long): k_sync_write's waits; then the launches behind it take over
struct SpinGuard {
    unsigned long long t0 = 0, limit;
    uint32_t polls = 0;
    __device__ __forceinline__ explicit SpinGuard(unsigned long long ticks) : limit(ticks) {}
    // called after a failed poll; true = give up
    __device__ __forceinline__ bool expired()
    {
        if (polls++ == 0) {
            t0 = __builtin_amdgcn_s_memrealtime();
            return false;
        }
        if (polls & 15) return false;   // a clock read every 16 polls
        return __builtin_amdgcn_s_memrealtime() - t0 > limit;
    }
};

#if KPEG_SYNC_STATS
// experiment builds: per-wavefront timeline of K1's pass 0 and of K2 ([wave][8] each; tools/sync_dbg.py)
__device__ unsigned long long g_ent_stamp[2][8192 * 16];
#endif

struct EntropyScratch {
    void* d_u = nullptr;        size_t u_cap = 0;       // un-stuffed words
    void* d_part = nullptr;     size_t part_cap = 0;    // K0's look-back words, one per 8 KiB of scan
    bool part_clean = false;    // d_part is all zero (the previous call's last K1 launch cleared what that call used)
    void* d_segoff = nullptr;   size_t seg_cap = 0;     // seg_off[S+1], sub_base[S+1]
    void* d_state = nullptr;    size_t state_cap = 0;   // X[nsub], Xb[2][nwg], assumed[nwg] uint64
    void* d_cnt = nullptr;      size_t cnt_cap = 0;     // cnt[nsub] int4
    void* d_wsum = nullptr;     size_t wsum_cap = 0;
    void* d_nrec = nullptr;     size_t nrec_cap = 0;    // nrec[nsub], wrec[nwg]: record counts of the compact coefficient stream
    void* d_flags = nullptr;    size_t flags_cap = 0;   // k_sync_write: pub[nwg][PUB_WORDS]; zero when (re)allocated
    uint32_t gen = 0;           // ... and the number of its last call (never 0 in a flag)
    EntropyMeta* d_meta = nullptr;
    EntropyTables* d_tabs = nullptr;
    EntropyTables h_tabs_cached;
    bool tabs_valid = false;
};

static void entropy_scratch_free(EntropyScratch* s)
{
    void* ps[] = {s->d_u, s->d_part, s->d_segoff, s->d_state, s->d_cnt, s->d_wsum, s->d_nrec, s->d_flags, s->d_meta, s->d_tabs};
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
    int warm;          // warm-up sub-sequences per workgroup, < 0 = default (test hook: 0 makes every workgroup guess wrong)
    int subseq = 0;    // sub-sequence size: 0 = chosen from the bit rate, else SUBSEQ_SPARSE or SUBSEQ_DENSE (test hook)
    // compact coefficient stream instead of the dense layout (d_coef unused): see WriteArgs
    uint32_t* d_rec = nullptr;
    uint32_t rec_cap = 0;
    int16_t* d_dc16 = nullptr;
    uint32_t* d_tile_start = nullptr;
    uint32_t ntiles = 0;
    unsigned long long spin_ticks = 0;   // bound of the waits between workgroups in 100 MHz ticks, 0 = defaults (test hook)
    uint32_t fault = 0;                  // fault injection (test hook): bit 0 K0's, bit 1 K1's workgroup 0 never publishes
    uint32_t gray = 0;                   // one-component stream (kpeg_frame::components == 1)
    uint32_t sub420 = 0;                 // 4:2:0 (extension, dense layout): nmcu counts 16x16 MCUs of six blocks
    uint32_t force_k0 = 0;               // test / experiment hook: run K0 even where K1 and K2 could un-stuff for themselves
    uint32_t fused_slots = 0;            // workgroups of k_sync_write the device holds at once (0: that kernel is not used)
    // fused batch: nimg > 0 independent scans decoded as the restart segments of one virtual stream (each starts
    // from the known state, DC predictors reset): d_scan / scan_len are unused, restart_interval = MCUs per image
    uint32_t nimg = 0;
    const uint8_t* const* d_scan_tab = nullptr;  // [nimg] device pointers (device array)
    const uint32_t* d_len_tab = nullptr;         // [nimg] scan lengths
    const uint32_t* d_wg_tab = nullptr;          // [nimg + 1] first K0 workgroup of each image
    uint32_t total_parts = 0;                    // K0 workgroups over all images
    uint64_t total_len = 0;                      // scan bytes over all images
};

// ------------------------------------------------------------------------------------------
// K0: unstuff
// keep(j): reference stream: drop b[j]==00 after b[j-1]==FF unless j is the last byte
// (byteStuffScanData's `i + 8 < size - 8`, Decoder.cpp:637).  With restart markers: drop every
// stuffed 00 and both bytes of FF D0..D7.
constexpr int US_BYTES_PER_THREAD = 16;
#ifndef KPEG_US_THREADS
#define KPEG_US_THREADS 512   // 8 KiB of scan per workgroup (256 / 512 / 1024 threads: K0 23.5 / 21.1 / 24.9 us on the 8K image)
#endif
constexpr int US_THREADS = KPEG_US_THREADS;
constexpr int US_BLOCK_BYTES = US_BYTES_PER_THREAD * US_THREADS;

// keep / marker flags of one thread's 16 bytes, held in w[0..3] (little-endian), prev = the byte before them,
// next = the byte after them (0x100 = none); nvalid = how many of the 16 exist
__device__ __forceinline__ void us_flags(const uint32_t w[4], uint32_t prev, uint32_t next, uint32_t nvalid, bool last_is_final, bool rst,
                                         uint32_t& keepmask, uint32_t& markmask)
{
    keepmask = 0;
    markmask = 0;
#pragma unroll
    for (int k = 0; k < US_BYTES_PER_THREAD; ++k) {
        const uint32_t cur = (w[k >> 2] >> ((k & 3) * 8)) & 0xFF;
        const uint32_t nxt = k + 1 < US_BYTES_PER_THREAD ? (w[(k + 1) >> 2] >> (((k + 1) & 3) * 8)) & 0xFF : next;
        const bool exists = (uint32_t)k < nvalid;
        const bool has_next = (uint32_t)(k + 1) < nvalid || (nvalid == US_BYTES_PER_THREAD && !last_is_final);
        bool keep = true;
        if (rst) {
            const uint32_t nx = has_next ? nxt : 0x100u;
            if (cur == 0x00 && prev == 0xFF) keep = false;
            if (cur == 0xFF && nx >= 0xD0 && nx <= 0xD7) {
                keep = false;
                if (exists) markmask |= 1u << k;  // a marker starts here
            }
            if (prev == 0xFF && cur >= 0xD0 && cur <= 0xD7) keep = false;
        } else {
            if (cur == 0x00 && prev == 0xFF && has_next) keep = false;  // never the last byte
        }
        if (keep && exists) keepmask |= 1u << k;
        prev = cur;
    }
}

// One launch: every workgroup flags its 8 KiB (one 16-byte load per thread), scans its keep/marker counts,
// gets its base from its predecessors by decoupled look-back (aggregate / inclusive prefix published in one
// 64-bit word: kept bytes [27:0], markers [54:28], state [63:62]; workgroups are dispatched in index order),
// compacts its kept bytes in LDS in their final word order and writes them out as whole words.  The last
// workgroup knows the totals; without restart markers it also does k_seg_setup's job (one segment), which
// saves that launch.  part[] must be zero on entry: K1's last launch clears it for the next call.
constexpr unsigned long long LB_AGG = 1ull << 62, LB_PFX = 2ull << 62;
// Restart segments (RSTn markers, or the images of a fused batch): seg_off[nseg] = n_u, bytes past the end
// zeroed, sub_base[] = exclusive scan of ceil(seg_bits / SUBSEQ_BITS), bookkeeping reset.  Run by the last
// workgroup of k_unstuff to finish (US_THREADS threads); seg_off[] was written by other workgroups of the
// same launch, hence the L2 loads.
__device__ void seg_setup_wg(EntropyMeta* meta, uint32_t* seg_off, uint32_t* sub_base, uint32_t seg_cap, uint32_t nsub_cap, uint8_t* u,
                             uint32_t n_u, uint32_t nseg, uint32_t expected_segs, uint32_t* status, uint32_t subseq_bits, uint32_t* s /* [US_THREADS] */,
                             uint32_t* carry)
{
    const uint32_t t = threadIdx.x;
    if (t == 0) {
        meta->n_u = n_u;
        meta->nseg = nseg;
        for (int i = 0; i < SYNC_PASSES + 8; ++i) meta->moved[i] = 0;
        meta->total_blocks = 0;
        meta->ticket = 0;
        meta->k1_order = 0;
    }
    if (nseg != expected_segs || nseg + 1 > seg_cap) {
        if (t == 0) {
            atomicOr(&status[1], 1u);  // restart markers do not match the restart interval
            status[4] = nseg;          // (diagnostics: what was counted, what the frame says)
            status[5] = expected_segs;
            meta->nsub = 0;
        }
        return;
    }
    if (t == 0) {
        seg_off[nseg] = n_u;
        *carry = 0;
    }
    if (t < 16) u[(n_u + t) ^ 3] = 0;   // bytes past the end read as zero (the readers look ahead)
    __syncthreads();
    auto seg_at = [&](uint32_t r) -> uint32_t {
        return r == nseg ? n_u : __hip_atomic_load(&seg_off[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    for (uint32_t base = 0; base < nseg; base += US_THREADS) {
        const uint32_t r = base + t;
        uint32_t v = 0;
        if (r < nseg) {
            const uint32_t bits = (seg_at(r + 1) - seg_at(r)) * 8;
            v = (bits + subseq_bits - 1) / subseq_bits;
            if (v == 0) v = 1;
        }
        s[t] = v;
        __syncthreads();
        for (int o = 1; o < US_THREADS; o <<= 1) {
            uint32_t x = 0;
            if ((int)t >= o) x = s[t - o];
            __syncthreads();
            s[t] += x;
            __syncthreads();
        }
        const uint32_t c = *carry;
        if (r < nseg) sub_base[r] = c + s[t] - v;
        __syncthreads();
        if (t == US_THREADS - 1) *carry = c + s[US_THREADS - 1];
        __syncthreads();
    }
    if (t == 0) {
        sub_base[nseg] = *carry;
        if (*carry > nsub_cap) {
            atomicOr(&status[1], 2u);
            meta->nsub = 0;
        } else {
            meta->nsub = *carry;
        }
    }
}

struct UnstuffBatch {   // nimg > 0: workgroup g belongs to the image whose [wg_tab[i], wg_tab[i+1]) holds g
    uint32_t nimg;
    const uint8_t* const* scan_tab;
    const uint32_t* len_tab;
    const uint32_t* wg_tab;
};
// the image a K0 workgroup belongs to (fused batch), its index inside that image, the image's bytes
__device__ __forceinline__ void us_locate(const UnstuffBatch& bt, uint32_t g, const uint8_t*& b, uint32_t& n, uint32_t& gl, uint32_t& img)
{
    gl = g;
    img = 0;
    if (bt.nimg) {
        uint32_t lo = 0, hi = bt.nimg;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (bt.wg_tab[mid] <= g) lo = mid;
            else hi = mid;
        }
        img = lo;
        gl = g - bt.wg_tab[img];
        b = bt.scan_tab[img];
        n = bt.len_tab[img];
    }
}

// Look-back fallback: (kept bytes, markers) of workgroup j's 8 KiB, computed by one wavefront from the input bytes --
// what workgroup j publishes as its aggregate.  Called with the whole wavefront converged.
__device__ __forceinline__ unsigned long long us_aggregate_wave(const UnstuffBatch& bt, uint32_t j, const uint8_t* b, uint32_t n, bool rst)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t gl, img;
    us_locate(bt, j, b, n, gl, img);
    uint32_t k = 0, m = 0;
    for (uint32_t s = 0; s < US_THREADS / 64; ++s) {
        const uint32_t j0 = (gl * US_THREADS + s * 64 + lane) * US_BYTES_PER_THREAD;
        const uint32_t nvalid = j0 >= n ? 0u : min((uint32_t)US_BYTES_PER_THREAD, n - j0);
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t q = 0; q < nvalid; ++q) w[q >> 2] |= (uint32_t)b[j0 + q] << ((q & 3) * 8);
        const uint32_t prev = j0 > 0 && j0 - 1 < n ? b[j0 - 1] : 0x100u;
        const uint32_t next = j0 + US_BYTES_PER_THREAD < n ? b[j0 + US_BYTES_PER_THREAD] : 0x100u;
        uint32_t km, mm;
        us_flags(w, prev, next, nvalid, j0 + US_BYTES_PER_THREAD >= n, rst, km, mm);
        k += __popc(km);
        m += __popc(mm);
    }
    for (int o = 32; o > 0; o >>= 1) {
        k += __shfl_xor(k, o);
        m += __shfl_xor(m, o);
    }
    return (unsigned long long)k | ((unsigned long long)m << 28);
}

__global__ __launch_bounds__(US_THREADS) void k_unstuff(const uint8_t* b, uint32_t n, int rst, unsigned long long* part, uint8_t* u,
                                                        uint32_t* seg_off, uint32_t seg_cap, EntropyMeta* meta, uint32_t* sub_base,
                                                        uint32_t nsub_cap, uint32_t* status, UnstuffBatch bt, uint32_t expected_segs,
                                                        uint32_t subseq_bits, unsigned long long spin_ticks, uint32_t fault)
{
    __shared__ uint32_t s_wave[US_THREADS / 64];
    __shared__ uint32_t s_base[2];
    __shared__ uint32_t s_dep, s_last;
    __shared__ uint32_t s_out[US_BLOCK_BYTES / 4 + 2];
    const uint32_t g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint8_t* const b0 = b;
    const uint32_t n0 = n;
    uint32_t gl = g, img = 0;   // workgroup index inside its image
    us_locate(bt, g, b, n, gl, img);
    const uint32_t j0 = (gl * US_THREADS + t) * US_BYTES_PER_THREAD;
    const uint32_t nvalid = j0 >= n ? 0u : min((uint32_t)US_BYTES_PER_THREAD, n - j0);
    uint32_t w[4] = {0, 0, 0, 0};
    if (nvalid == US_BYTES_PER_THREAD && (reinterpret_cast<uintptr_t>(b) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(b + j0);
        w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
    } else {
        for (uint32_t k = 0; k < nvalid; ++k) w[k >> 2] |= (uint32_t)b[j0 + k] << ((k & 3) * 8);
    }
    // the bytes around: from the neighbouring lanes, from memory at the wavefront's edges
    uint32_t prev = (uint32_t)__shfl_up((int)(w[3] >> 24), 1), next = (uint32_t)__shfl_down((int)(w[0] & 0xFF), 1);
    if (lane == 0) prev = j0 > 0 && j0 - 1 < n ? b[j0 - 1] : 0x100u;
    if (lane == 63) next = j0 + US_BYTES_PER_THREAD < n ? b[j0 + US_BYTES_PER_THREAD] : 0x100u;
    uint32_t km, mm;
    us_flags(w, prev, next, nvalid, j0 + US_BYTES_PER_THREAD >= n, rst != 0, km, mm);
    // inclusive scan of (kept, markers) packed as kept | markers << 16: inside the wavefront, then over the four wavefronts
    const uint32_t own = __popc(km) | (__popc(mm) << 16);
    uint32_t v = own;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = (uint32_t)__shfl_up((int)v, o);
        if ((int)lane >= o) v += x;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (uint32_t q = 0; q < US_THREADS / 64; ++q) {
        if (q < wave) woff += s_wave[q];
        total += s_wave[q];
    }
    const uint32_t tk = total & 0xFFFF, tm = total >> 16;
    if (t < 64) {
        // wavefront 0 looks back 64 predecessors at a time: the nearest inclusive prefix ends the walk
        const unsigned long long mine = (unsigned long long)tk | ((unsigned long long)tm << 28);
        const bool mute = (fault & 1u) && g == 0 && gridDim.x > 1;   // test hook: workgroup 0 never publishes, its successors fall back
        // (test hook, fault bit 2: nobody publishes an aggregate ahead of its prefix -- with a short bound the look-back
        // then computes many predecessors' aggregates itself, whatever kind of chunk they are)
        if (lane == 0 && g > 0 && !(fault & 4u)) __hip_atomic_store(&part[g], mine | LB_AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t acc_k = 0, acc_m = 0;
        for (int base = (int)g - 1; base >= 0; base -= 64) {
            const int j = base - (int)lane;
            unsigned long long pv = j >= 0 ? 0ull : LB_PFX;   // before the first workgroup: prefix 0
            // Poll the 64 predecessors together.  One that has published nothing within the bound (not dispatched yet, or
            // whatever else) gets its aggregate computed here from its input bytes -- the nearest such predecessor only,
            // then the poll goes on: a merely slow neighbourhood costs one fallback per bound, not sixty-four.
            SpinGuard guard(spin_ticks);
            for (;;) {
                if (j >= 0 && (pv >> 62) == 0) pv = __hip_atomic_load(&part[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long pending = __ballot((pv >> 62) == 0);
                if (!pending) break;
                if (!guard.expired()) {   // (every lane polls in step: the guard's state is wave-uniform)
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                const int L = __builtin_ctzll(pending);
                const uint32_t jL = (uint32_t)__builtin_amdgcn_readlane(j, L);
                const unsigned long long agg = us_aggregate_wave(bt, jL, b0, n0, rst != 0);
                if ((int)lane == L) pv = agg | LB_AGG;
                guard = SpinGuard(spin_ticks);
            }
            const unsigned long long pfx = __ballot((pv >> 62) == 2);
            const uint32_t first = pfx ? (uint32_t)__builtin_ctzll(pfx) : 63u;
            uint32_t k = lane <= first ? (uint32_t)(pv & 0xFFFFFFFu) : 0u;
            uint32_t m = lane <= first ? (uint32_t)((pv >> 28) & 0x7FFFFFFu) : 0u;
            for (int o = 32; o > 0; o >>= 1) {
                k += __shfl_down(k, o);
                m += __shfl_down(m, o);
            }
            acc_k += k;
            acc_m += m;
            if (pfx) break;
        }
        if (lane == 0) {
            const unsigned long long acc = (unsigned long long)acc_k | ((unsigned long long)acc_m << 28);
            // (an exchange: its return means it has been performed, which the ticket at the end relies on)
            s_dep = mute ? 0u : (uint32_t)atomicExch(&part[g], (acc + mine) | LB_PFX);
            s_base[0] = acc_k;
            s_base[1] = acc_m;
        }
    }
    __syncthreads();
    const uint32_t base_k = s_base[0], abase = base_k & ~3u;
    uint32_t dep = 0;   // values returned by this thread's segment-offset exchanges
    {
        const uint32_t excl = woff + v - own;
        uint32_t pos = base_k + (excl & 0xFFFF);
        uint32_t mk = s_base[1] + (excl >> 16);
        uint8_t* so = reinterpret_cast<uint8_t*>(s_out);
#pragma unroll
        for (int k = 0; k < US_BYTES_PER_THREAD; ++k) {
            if (mm & (1u << k)) {
                mk++;
                if (mk < seg_cap) dep |= atomicExch(&seg_off[mk], pos);  // segment mk starts at the next kept byte
            }
            if (km & (1u << k)) {
                so[(pos - abase) ^ 3] = (uint8_t)(w[k >> 2] >> ((k & 3) * 8));  // big-endian words for little-endian 32-bit loads
                pos++;
            }
        }
    }
    __syncthreads();
    {
        // this workgroup's bytes [base_k, base_k + tk): whole words as words, the shared edge words byte by byte
        const uint32_t end_k = base_k + tk;
        const uint32_t nwords = tk ? ((end_k + 3) >> 2) - (abase >> 2) : 0u;
        uint32_t* uw = reinterpret_cast<uint32_t*>(u) + (abase >> 2);
        const uint8_t* so = reinterpret_cast<const uint8_t*>(s_out);
        for (uint32_t wi = t; wi < nwords; wi += US_THREADS) {
            const uint32_t p0 = abase + 4 * wi;
            if (p0 >= base_k && p0 + 4 <= end_k) {
                uw[wi] = s_out[wi];
            } else {
                for (uint32_t p = max(p0, base_k); p < min(p0 + 4, end_k); ++p) u[p ^ 3] = so[(p - abase) ^ 3];
            }
        }
    }
    if (g == 0 && t == 0) dep |= atomicExch(&seg_off[0], 0u);
    if (bt.nimg && gl == 0 && t == 0 && img < seg_cap) dep |= atomicExch(&seg_off[img], base_k);   // image img = segment img of the virtual stream
    if (!rst && !bt.nimg) {
        if (g == gridDim.x - 1) {
            // one segment, and this workgroup knows the total: the set-up in place
            const uint32_t n_u = base_k + tk;
            if (t < 16) u[(n_u + t) ^ 3] = 0;   // bytes past the end read as zero (the readers look ahead)
            if (t == 0) {
                meta->n_u = n_u;
                meta->nseg = 1;
                uint32_t nsub = (n_u * 8 + subseq_bits - 1) / subseq_bits;
                if (nsub == 0) nsub = 1;
                seg_off[1] = n_u;
                sub_base[0] = 0;
                sub_base[1] = nsub;
                if (nsub > nsub_cap) {
                    atomicOr(&status[1], 2u);
                    nsub = 0;
                }
                meta->nsub = nsub;
                for (int i = 0; i < SYNC_PASSES + 8; ++i) meta->moved[i] = 0;
                meta->total_blocks = 0;
                meta->ticket = 0;
                meta->k1_order = 0;
            }
        }
        return;
    }
    // Restart segments: their offsets come from many workgroups, so the last one to finish sets them up.
    // Two-level ticket (a thousand tickets on one word queue up at the kernel's tail); no fences (a release
    // fence writes back the whole L2): every write the set-up reads is an exchange whose return this
    // workgroup has waited for.
    if (dep == 0xDEADBEEFu && t == 1) s_dep ^= dep;   // (keeps the returned values live: the exchanges are waited for)
    __syncthreads();
    if (t == 0) {
        const uint32_t slot = g & 63, in_slot = (gridDim.x - slot + 63) >> 6;
        uint32_t last = 0;
        if (atomicAdd(&meta->k0_slot[slot], 1u + (s_dep & 0u)) == in_slot - 1)
            last = atomicAdd(&meta->k0_top, 1u) == min(gridDim.x, 64u) - 1 ? 1u : 0u;
        s_last = last;
    }
    __syncthreads();
    if (s_last) {
        const unsigned long long tot = __hip_atomic_load(&part[gridDim.x - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t n_u = (uint32_t)(tot & 0xFFFFFFFu), nseg = bt.nimg ? bt.nimg : (uint32_t)((tot >> 28) & 0x7FFFFFFu) + 1;
        seg_setup_wg(meta, seg_off, sub_base, seg_cap, nsub_cap, u, n_u, nseg, expected_segs, status, subseq_bits, s_out, &s_base[0]);
        if (t < 64) meta->k0_slot[t] = 0;
        if (t == 64) meta->k0_top = 0;
    }
}

// ------------------------------------------------------------------------------------------
// Decoder core

struct DecState {   // at a codeword boundary
    uint32_t p;     // bit position in the un-stuffed string
    uint32_t c;     // component 0..2 of the block being decoded (4:2:0: block 0..5 of the MCU, Y Y Y Y Cb Cr)
    uint32_t k;     // 0: next symbol is the DC symbol; 1..63: AC, k-1 coefficients placed so far
    uint32_t q;     // k != 0: 1 if the block in progress keeps its AC terms (its DC symbol was not 0x00, quirk Q1); else 0
};
__device__ __forceinline__ uint64_t pack_state(const DecState& s)
{
    return (uint64_t)s.p | ((uint64_t)s.c << 32) | ((uint64_t)s.k << 35) | ((uint64_t)s.q << 42);
}
__device__ __forceinline__ DecState unpack_state(uint64_t v)
{
    DecState s;
    s.p = (uint32_t)v;
    s.c = (uint32_t)(v >> 32) & 7;
    s.k = (uint32_t)(v >> 35) & 127;
    s.q = (uint32_t)(v >> 42) & 1;
    return s;
}

// LDS image of the tables.  The first-level tables stand in decode order
//     DC comp 0, AC comp 0, DC comp 1, AC comp 1, DC comp 2, AC comp 2
// (components 1 and 2 share their Huffman tables: two copies), so the byte offset tb of the table
// in use is the decoder's whole (component, DC/AC) state: when a table's turn ends (a DC symbol, an
// EOB, the 63rd AC coefficient) tb moves on by one table and wraps after AC comp 2.
constexpr uint32_t LUT_BYTES = (1u << LUT_BITS) * 4;
struct LdsTables {
    uint32_t lut[6][1 << LUT_BITS];
    uint32_t pool[POOL_SUBS][1 << LUT2_BITS];
    int32_t maxcode[4][18];
    int32_t valoff[4][18];
    uint8_t symbols[4][256];
    uint8_t zz[64];
    float mscale_zz[2][64];
    float2 zzm[2][64];
    float q00[2];
    float pad16[2];
};
static_assert(sizeof(LdsTables) - sizeof(uint32_t) * 6 * (1 << LUT_BITS) == offsetof(EntropyTables, lutx) - sizeof(uint32_t) * 4 * (1 << LUT_BITS),
              "layout after the first-level tables");

__device__ __forceinline__ uint32_t slot_table(uint32_t slot) { return (slot & 1) * 2 + (slot >= 2 ? 1 : 0); }  // [class*2+id]

__device__ __forceinline__ void load_tables(LdsTables* dst, const EntropyTables* src)
{
    // 16 bytes per load: 26 KB per workgroup, a few loads per thread (hipMalloc'd source, 16-byte aligned members)
    constexpr uint32_t LUT_V4 = (1u << LUT_BITS) / 4;   // uint4 per first-level table
    uint4* l = reinterpret_cast<uint4*>(&dst->lut[0][0]);
    for (uint32_t i = threadIdx.x; i < 6 * LUT_V4; i += blockDim.x) {
        const uint32_t slot = i / LUT_V4;
        l[i] = reinterpret_cast<const uint4*>(&src->lut[slot_table(slot)][0])[i - slot * LUT_V4];
    }
    constexpr uint32_t nbytes = offsetof(EntropyTables, lutx) - sizeof(uint32_t) * 4 * (1 << LUT_BITS);
    static_assert(nbytes % 16 == 0 && (sizeof(uint32_t) * 4 * (1 << LUT_BITS)) % 16 == 0 && (sizeof(uint32_t) * 6 * (1 << LUT_BITS)) % 16 == 0,
                  "tables are copied in 16-byte pieces");
    const uint4* s = reinterpret_cast<const uint4*>(&src->pool[0][0]);
    uint4* d = reinterpret_cast<uint4*>(&dst->pool[0][0]);
    for (uint32_t i = threadIdx.x; i < nbytes / 16; i += blockDim.x) d[i] = s[i];
}
// K1: the six slots once more, with the two-symbol entries.  DCPAIR = false (dense streams: a DC symbol there rarely has
// the next code inside the window, and the step is two instructions shorter without): the DC slots as they are.
template <bool DCPAIR>
__device__ __forceinline__ void load_tables_x(uint32_t* dst, const EntropyTables* src)
{
    constexpr uint32_t LUT_V4 = (1u << LUT_BITS) / 4;
    uint4* l = reinterpret_cast<uint4*>(dst);
    for (uint32_t i = threadIdx.x; i < 6 * LUT_V4; i += blockDim.x) {
        const uint32_t slot = i / LUT_V4;
        const uint32_t* tab = (DCPAIR || (slot & 1)) ? &src->lutx[slot_table(slot)][0] : &src->lut[slot_table(slot)][0];
        l[i] = reinterpret_cast<const uint4*>(tab)[i - slot * LUT_V4];
    }
}

// The bit string as the decode loops see it: the workgroup's slice staged in LDS.  Every position
// a lane can reach lies inside the slice: a run ends at most one symbol (< 32 bits) past its own
// sub-sequence and K2 finishes at most one block (< 63 * 31 bits) past it, plus 96 bits of
// look-ahead; STAGE_MARGIN words cover that.  Only a corrupt stream can run further (it reads
// whatever LDS holds there; the result is flagged as an error elsewhere).
constexpr int STAGE_MARGIN = 96;
__device__ __forceinline__ void stage_bits(uint32_t* lds, uint32_t cap, const uint32_t* __restrict__ g, uint32_t w0, uint32_t nw)
{
    for (uint32_t j = threadIdx.x; j < cap; j += blockDim.x) lds[j] = j < nw ? g[w0 + j] : 0u;
}

// ------------------------------------------------------------------------------------------
// Streams without restart markers, one image: no K0.  The sub-sequences are then chunks of S / 8 bytes of the scan AS IT IS
// (still byte-stuffed), and every workgroup of K1 and K2 un-stuffs the chunks it stages while it stages them: a chunk
// becomes 8 * (its kept bytes) bits of the workgroup's slice in LDS, start[] says where.  Bit positions that leave a
// workgroup (exit states) are VIRTUAL: chunk << VSHIFT | bit inside the chunk's kept bits (a decode ends less than a symbol,
// < 32 bits, past its chunk, and a chunk keeps at least half its bytes: the position lies inside the next chunk); the same
// stream position has the same virtual position in every workgroup.  byteStuffScanData's rule (Decoder.cpp:631-650): a
// 00 after an FF goes, except as the very last byte of the scan.
template <int S>
struct StuffedGeom {
    static constexpr uint32_t BYTES = S / 8;
    static constexpr uint32_t VSHIFT = S <= 96 ? 7 : 9;
    static constexpr uint32_t EXTRA = STAGE_MARGIN * 4 / BYTES;   // chunks staged past the workgroup's last one: the decoders' look-ahead
};

// Stages chunks [c0, c0 + nst) (nst <= SYNC_WG + EXTRA) of scan[0, n): lds receives the kept bytes as big-endian words from
// bit 0 (what the K0 path stages), start[j] the first bit of chunk c0 + j (start[nst] = the end).  red: SYNC_WG / 64 + 1
// words of scratch.  One round: thread t takes chunk t and, if t < nst - SYNC_WG, also chunk SYNC_WG + t (the few staged
// for the decoders' look-ahead); all loads are issued before anything waits for one.
template <int S>
struct StuffedChunk {
    static constexpr uint32_t BYTES = S / 8, WORDS = BYTES / 4;
    uint32_t w[WORDS];    // the chunk's bytes, little-endian words
    uint32_t drop[WORDS]; // bit 7 of every byte that goes: a 00 after an FF that is not the scan's last byte
    uint32_t nb, nk;      // bytes that exist, bytes kept

    // (only words that hold at least one byte of the scan are touched: an aligned word never straddles the end of a mapping,
    // so nothing outside the caller's buffer is read that does not share a word with it)
    __device__ __forceinline__ void load(const uint8_t* __restrict__ scan, uint32_t n, uint64_t b0, bool have, uint32_t& prev)
    {
        nb = have ? (uint32_t)min((uint64_t)BYTES, (uint64_t)n - b0) : 0u;
        prev = 0u;
#pragma unroll
        for (uint32_t q = 0; q < WORDS; ++q) w[q] = 0u;
        if (have) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(scan) + b0;
            const uint32_t* ap = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
            const uint32_t sh = (uint32_t)(a & 3);
            const uintptr_t lo = reinterpret_cast<uintptr_t>(scan) & ~(uintptr_t)3, hi = (reinterpret_cast<uintptr_t>(scan) + n + 3) & ~(uintptr_t)3;
            uint32_t raw[WORDS + 1];
#pragma unroll
            for (uint32_t q = 0; q < WORDS + 1; ++q) {
                const uintptr_t wa = reinterpret_cast<uintptr_t>(ap + q);
                raw[q] = (wa >= lo && wa < hi) ? ap[q] : 0u;
            }
#pragma unroll
            for (uint32_t q = 0; q < WORDS; ++q) w[q] = __builtin_amdgcn_alignbyte(raw[q + 1], raw[q], sh);
            if (b0 > 0) prev = scan[b0 - 1];
        }
    }
    __device__ __forceinline__ void flags(uint32_t n, uint64_t b0, uint32_t prev)
    {
        uint32_t dropped = 0;
#pragma unroll
        for (uint32_t q = 0; q < WORDS; ++q) {
            const uint32_t before = (w[q] << 8) | (q ? w[q - 1] >> 24 : prev);      // every byte's predecessor
            const uint32_t z = ~(((w[q] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w[q]) & 0x80808080u;            // bytes that are 00
            const uint32_t nf = ~before;
            const uint32_t f = ~(((nf & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | nf) & 0x80808080u;               // predecessors that are FF
            uint32_t d = z & f;
            // bytes that do not exist, and the scan's last byte, stay out of it
            const uint32_t first = q * 4;
            const uint32_t lim = b0 + nb >= n ? (nb ? nb - 1 : 0u) : nb;   // bytes [0, lim) may go
            const uint32_t keepbytes = lim > first ? min(lim - first, 4u) : 0u;
            d &= keepbytes >= 4 ? 0xFFFFFFFFu : ((1u << (8 * keepbytes)) - 1u);
            drop[q] = d;
            dropped += __popc(d);
        }
        nk = nb - dropped;
    }
    // the kept bytes to lds bytes [pos, pos + nk), big-endian words
    __device__ __forceinline__ void scatter(uint8_t* lb, uint32_t pos) const
    {
#pragma unroll
        for (uint32_t k = 0; k < BYTES; ++k) {
            const bool keep = k < nb && !((drop[k >> 2] >> ((k & 3) * 8 + 7)) & 1u);
            if (keep) {
                lb[pos ^ 3u] = (uint8_t)(w[k >> 2] >> ((k & 3) * 8));
                pos++;
            }
        }
    }
};

// In two steps, so that the caller can put its other global loads (the tables) between the chunks' loads and their use.
template <int S>
struct StuffedStage {
    StuffedChunk<S> A, B;
    uint32_t prevA, prevB, n1, n2, c0;

    __device__ __forceinline__ void begin(const uint8_t* __restrict__ scan, uint32_t n, uint32_t c0_, uint32_t nst)
    {
        constexpr uint32_t BYTES = StuffedGeom<S>::BYTES;
        const uint32_t t = threadIdx.x;
        c0 = c0_;
        n1 = min(nst, (uint32_t)SYNC_WG);   // chunks of the first kind (one per thread) ...
        n2 = nst - n1;                      // ... and of the second (the look-ahead's, wavefront 0 only)
        const uint64_t bA = (uint64_t)(c0 + t) * BYTES;
        A.load(scan, n, bA, t < n1 && bA < n, prevA);
        B.nb = B.nk = 0;
        prevB = 0;
        if (t < 64) {   // (wave-uniform)
            const uint64_t bB = (uint64_t)(c0 + SYNC_WG + t) * BYTES;
            B.load(scan, n, bB, t < n2 && bB < n, prevB);
        }
    }
    __device__ __forceinline__ void finish(uint32_t* lds, uint32_t cap, uint32_t* start, uint32_t* red, uint32_t n)
    {
        constexpr uint32_t BYTES = StuffedGeom<S>::BYTES;
        const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
        for (uint32_t j = t; j < cap; j += SYNC_WG) lds[j] = 0u;   // look-ahead past the end reads zero bits
        A.flags(n, (uint64_t)(c0 + t) * BYTES, prevA);
        uint32_t incA = A.nk, incB = 0;
        if (t < 64) {
            B.flags(n, (uint64_t)(c0 + SYNC_WG + t) * BYTES, prevB);
            incB = B.nk;
        }
        // exclusive scans of the kept counts: the first kind over the workgroup, the second inside wavefront 0
        incA = wave_scan_incl(incA);
        if (t < 64) incB = wave_scan_incl(incB);   // (wave-uniform)
        if (lane == 63) red[wave] = incA;
        __syncthreads();   // (also: the zero fill is done)
        uint32_t base = 0, tot = 0;
        for (uint32_t q = 0; q < SYNC_WG / 64; ++q) {
            const uint32_t x = red[q];
            if (q < wave) base += x;
            tot += x;
        }
        uint8_t* lb = reinterpret_cast<uint8_t*>(lds);
        const uint32_t posA = base + incA - A.nk, posB = tot + incB - B.nk;
        if (t <= n1 && (t < n1 || n2 == 0)) start[t] = posA * 8;          // (t == n1 with no second kind: the end)
        if (n2 && t <= n2 && t < 64) start[SYNC_WG + t] = posB * 8;        // (EXTRA <= 64; t == n2: the end)
        if (t == 0 && n2 == 0 && n1 == (uint32_t)SYNC_WG) start[n1] = tot * 8;   // (the thread that would write the end does not exist)
        A.scatter(lb, posA);
        if (t < 64 && t < n2) B.scatter(lb, posB);
        __syncthreads();
    }
};

// Sequential reader: the next >= 32 bits sit MSB-aligned in a 64-bit register, so a symbol's
// critical path is one LUT read, not LUT + two word fetches; the following word is fetched one
// refill ahead.
struct BitReader {
    uint64_t buf;
    uint32_t have;        // valid bits in buf, > 32 between symbols
    const uint32_t* wp;   // word after `nextw` in the staged slice
    uint32_t nextw;

    __device__ __forceinline__ void init(const uint32_t* lds, uint32_t w0, uint32_t p)
    {
        const uint32_t o = p & 31;
        wp = lds + ((p >> 5) - w0);
        buf = (((uint64_t)wp[0] << 32) | wp[1]) << o;
        have = 64 - o;
        nextw = wp[2];
        wp += 3;
    }
    __device__ __forceinline__ uint32_t peek() const { return (uint32_t)(buf >> 32); }
    __device__ __forceinline__ void consume(uint32_t n)   // n <= 31
    {
        buf <<= n;
        have -= n;
        if (have <= 32) {
            buf |= (uint64_t)nextw << (32 - have);
            have += 32;
            nextw = *wp++;
        }
    }
};

// canonical search, for tables with more long-code prefixes than the pool holds (never for real tables)
__device__ __attribute__((noinline)) uint32_t search_entry(const LdsTables& T, uint32_t tb, uint32_t win)
{
    const uint32_t slot = tb / LUT_BYTES, ti = slot_table(slot);
    const bool isdc = !(slot & 1);
    for (int l = LUT_BITS + 1; l <= 16; ++l) {
        const int code = (int)(win >> (32 - l));
        if (code <= T.maxcode[ti][l]) return make_entry((uint32_t)l, T.symbols[ti][(T.valoff[ti][l] + code) & 255], isdc);
    }
    return bad_entry(isdc);
}

// The entry of the symbol at the head of `win` (the next 32 bits), table at byte offset tb, in two
// steps so that a decode loop can issue the first-level read of the next symbol before it finishes
// the bookkeeping of the current one (the LDS latency is the longest link of the symbol-to-symbol chain).
__device__ __forceinline__ uint32_t lut_first(const LdsTables& T, uint32_t tb, uint32_t win)
{
    const char* base = reinterpret_cast<const char*>(&T.lut[0][0]);
    return *reinterpret_cast<const uint32_t*>(base + tb + ((win >> (32 - LUT_BITS)) << 2));
}
__device__ __forceinline__ uint32_t lut_finish(const LdsTables& T, uint32_t tb, uint32_t win, uint32_t e)
{
    if (e & E_LONG) {  // 10..16-bit code: one more table read
        const uint32_t sub = e & 0xFFFFu;
        e = sub != E_SEARCH ? T.pool[sub][(win >> (32 - 16)) & ((1u << LUT2_BITS) - 1)] : search_entry(T, tb, win);
    }
    return e;
}

// JPEG EXTEND (bitStringtoValue, Image.cpp:285-302) of the `cat` bits that follow a `len`-bit code
__device__ __forceinline__ int extend_win(uint32_t win, uint32_t len, uint32_t cat)
{
    const uint32_t t = win << len;  // magnitude bits at the top
    const uint32_t bits = __builtin_amdgcn_ubfe(t, 32 - cat, cat);   // 0 when cat == 0
    const int ones = (int)((1u << cat) - 1u);
    return (int)bits - ((int)t < 0 ? 0 : ones);
}

__device__ __forceinline__ uint32_t state_table(const DecState& s) { return (s.c * 2 + (s.k != 0 ? 1u : 0u)) * LUT_BYTES; }

struct RunResult {
    uint64_t exit_state;
    int4 cnt;     // blocks started (DC symbols decoded), sums of their DC differences per component
    uint32_t nrec;   // non-zero AC coefficients K2 will emit for this run (records of the compact coefficient stream)
#if KPEG_SYNC_STATS
    uint32_t iters;
#endif
};

// Sync/count run: decode from `s` until the bit position reaches `pend`.  COUNT: also count the records of the compact
// coefficient stream (left out where the dense layout is written: K1 is the pipeline's longest kernel).
// gray (wave-uniform): one component -- the table sequence is DC0 AC0 and every block adds to the one DC sum.
// SUMS = false: only the exit state is wanted (the first decode of K1, from a guessed entry state: whatever it counts is
// thrown away with the guess) -- the DC symbols then cost no more than any other.
// S420 (extension, dense layout): MCUs of six blocks Y Y Y Y Cb Cr.  tb then counts twelve VIRTUAL table slots (DC AC per
// block); the six tables in LDS serve them (phys_table), and the DC sums are kept per component instead of rotating.
template <bool S420>
__device__ __forceinline__ uint32_t phys_table(uint32_t tb)
{
    return S420 ? ((tb & LUT_BYTES) | (tb >= 8 * LUT_BYTES ? 2 * LUT_BYTES : 0u)) : tb;
}
template <bool COUNT, bool SUMS = true, bool S420 = false>
__device__ __forceinline__ RunResult run_count(const LdsTables& T, const uint32_t* bits, uint32_t w0, DecState s, uint32_t pend, bool gray)
{
    const uint32_t tb_wrap = S420 ? 12 * LUT_BYTES : (gray ? 2 * LUT_BYTES : 6 * LUT_BYTES);
    BitReader br;
    br.init(bits, w0, s.p);
    uint32_t p = s.p, k = s.k, q = s.q, tb = state_table(s);
    const uint32_t cfirst = s.k == 0 ? s.c : (s.c == 2 ? 0u : s.c + 1);  // component of the first block started here
    // DC sums rotate with the blocks: the current block's component adds into the slot that moves to the back
    int s0 = 0, s1 = 0, s2 = 0;
    uint32_t nb = 0, nrec = 0;
#if KPEG_SYNC_STATS
    uint32_t iters = 0;
#endif
    uint32_t e1 = lut_first(T, phys_table<S420>(tb), br.peek());
    while (p < pend) {
        const uint32_t win = br.peek();
        const uint32_t tbo = tb;
        const uint32_t e = lut_finish(T, phys_table<S420>(tb), win, e1);
        const uint32_t kraw = k + ((e >> 16) & 127);
        const bool adv = kraw >= 64;   // this table's turn ends: DC symbol, EOB, 63rd coefficient (Decoder.cpp:759)
        // exactly K2's condition for storing an AC coefficient, minus its check that the block lies inside the segment
        // (K2 never emits more records than are counted here: the counts fix where every lane's records go)
        if (COUNT && SUMS) nrec += ((e >> 26) & 1u) & q & (kraw <= 64 ? 1u : 0u);
        k = adv ? ((e >> 14) & 1u) : kraw;   // after a DC symbol 1, after a block 0
        q = adv ? ((e >> 25) & 1u) : q;
        tb += adv ? LUT_BYTES : 0u;
        tb = tb == tb_wrap ? 0u : tb;
        p += e & 31;
        br.consume(e & 31);
        e1 = lut_first(T, phys_table<S420>(tb), br.peek());   // next symbol's entry on its way (one read too many at the end: harmless)
        if (SUMS && (e & E_ISDC)) {
            const int d = extend_win(win, (e >> 5) & 31, (e >> 10) & 15);
            if (S420) {
                s0 += tbo < 8 * LUT_BYTES ? d : 0;
                s1 += tbo == 8 * LUT_BYTES ? d : 0;
                s2 += tbo == 10 * LUT_BYTES ? d : 0;
            } else {
                const int n = s0 + d;
                s0 = gray ? n : s1;
                s1 = gray ? s1 : s2;
                s2 = gray ? s2 : n;
            }
            nb++;
        }
#if KPEG_SYNC_STATS
        iters++;
#endif
    }
    RunResult r;
    // slot j holds component (cfirst + nb + j) mod 3
    const uint32_t rot = (gray || S420) ? 0u : (cfirst + nb) % 3;
    r.nrec = nrec;
    r.cnt.x = (int)nb;
    r.cnt.y = rot == 0 ? s0 : (rot == 1 ? s2 : s1);
    r.cnt.z = rot == 0 ? s1 : (rot == 1 ? s0 : s2);
    r.cnt.w = rot == 0 ? s2 : (rot == 1 ? s1 : s0);
    DecState x;
    x.p = p;
    x.c = tb / (2 * LUT_BYTES);
    x.k = k;
    x.q = q;
    r.exit_state = pack_state(x);
#if KPEG_SYNC_STATS
    r.iters = iters;
#endif
    return r;
}

// The exit state alone, as fast as it can be had: K1's first decode (from a guessed state) and the re-decodes of its rounds,
// whose chains -- one lane after the other -- are what the kernel's duration is made of.  While the sub-sequence's end is
// more than a first symbol (< LUT_BITS bits) away, symbols go two per step where the table has them (EntropyTables::lutx:
// AC AC below k = 48 only, the single-symbol tables take over from there; DC AC; DC EOB); the last symbols go one by one, so
// that the run stops at the FIRST symbol boundary at or past `pend`, like every other decode of the same sub-sequence.
template <bool DCPAIR>
__device__ __forceinline__ uint64_t run_exit(const LdsTables& T, const uint32_t* lutx, const uint32_t* bits, uint32_t w0, DecState s, uint32_t pend,
                                             bool gray
#if KPEG_SYNC_STATS
                                             , uint32_t* iters_out
#endif
)
{
    const uint32_t tb_wrap = gray ? 2 * LUT_BYTES : 6 * LUT_BYTES;
    BitReader br;
    br.init(bits, w0, s.p);
    uint32_t p = s.p, k = s.k, q = s.q, tb = state_table(s);
#if KPEG_SYNC_STATS
    uint32_t iters = 0;
#endif
    const char* const base_s = reinterpret_cast<const char*>(&T.lut[0][0]);
    const char* const base_x = reinterpret_cast<const char*>(lutx);
    constexpr uint32_t GUARD = LUT_BITS - 1;   // a first symbol of a pair is at most this long
    if (p + GUARD < pend) {
        uint32_t e1 = *reinterpret_cast<const uint32_t*>((k >= 48 ? base_s : base_x) + tb + ((br.peek() >> (32 - LUT_BITS)) << 2));
        do {
            const uint32_t win = br.peek();
            uint32_t e = e1;
            if (e & E_LONG) e = lut_finish(T, tb, win, e) & ~E_DCRUN;   // (a one-symbol entry: its bit 24 means something else)
            const uint32_t kraw = k + ((e >> 16) & 127);
            const bool adv = kraw >= 64;
            k = adv ? ((e >> 28) & 7u) : kraw;
            q = adv ? ((e >> 25) & 1u) : q;
            tb += adv ? (DCPAIR ? LUT_BYTES << ((e >> 24) & 1u) : LUT_BYTES) : 0u;   // (a DC symbol with the EOB behind it ends the turn of two tables)
            tb = tb == tb_wrap ? 0u : tb;
            p += e & 31;
            br.consume(e & 31);
            e1 = *reinterpret_cast<const uint32_t*>((k >= 48 ? base_s : base_x) + tb + ((br.peek() >> (32 - LUT_BITS)) << 2));
#if KPEG_SYNC_STATS
            iters++;
#endif
        } while (p + GUARD < pend);
    }
    uint32_t e1 = lut_first(T, tb, br.peek());
    while (p < pend) {
        const uint32_t win = br.peek();
        const uint32_t e = lut_finish(T, tb, win, e1);
        const uint32_t kraw = k + ((e >> 16) & 127);
        const bool adv = kraw >= 64;
        k = adv ? ((e >> 14) & 1u) : kraw;
        q = adv ? ((e >> 25) & 1u) : q;
        tb += adv ? LUT_BYTES : 0u;
        tb = tb == tb_wrap ? 0u : tb;
        p += e & 31;
        br.consume(e & 31);
        e1 = lut_first(T, tb, br.peek());
#if KPEG_SYNC_STATS
        iters++;
#endif
    }
#if KPEG_SYNC_STATS
    *iters_out = iters;
#endif
    DecState x;
    x.p = p;
    x.c = tb / (2 * LUT_BYTES);
    x.k = k;
    x.q = q;
    return pack_state(x);
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
// n_u: the un-stuffed length (meta->n_u).  One segment (a stream without restart markers) needs none of the tables:
// seg_off = {0, n_u}, sub_base = {0, ...} -- and no load that waits for another.
template <int S>
__device__ __forceinline__ SubGeom sub_geom(const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ sub_base,
                                            uint32_t nseg, uint32_t n_u, uint32_t i)
{
    KPEG_GEOMETRY(S);
    SubGeom g;
    if (nseg == 1) {
        g.seg = 0;
        g.li = i;
        const uint32_t s1 = n_u * 8;
        g.pstart = min(i * SUBSEQ_BITS, s1);
        g.pend = min(i * SUBSEQ_BITS + SUBSEQ_BITS, s1);
        return g;
    }
    g.seg = nseg > 1 ? locate_segment(sub_base, nseg, i) : 0;
    g.li = i - sub_base[g.seg];
    uint32_t s0 = seg_off[g.seg] * 8, s1 = seg_off[g.seg + 1] * 8;
    g.pstart = s0 + g.li * SUBSEQ_BITS;
    g.pend = min(g.pstart + SUBSEQ_BITS, s1);
    if (g.pstart > s1) g.pstart = s1;
    return g;
}

__device__ __forceinline__ int4 add4(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ------------------------------------------------------------------------------------------
// K1: sync
struct SyncArgs {
    const uint32_t* u;
    const uint32_t* seg_off;
    const uint32_t* sub_base;
    EntropyMeta* meta;
    const EntropyTables* tabs;
    uint64_t* X;        // [nsub_cap] exit state of every sub-sequence (written by its workgroup only)
    uint64_t* Xb;       // [2][nwg_cap] exit state of each workgroup's last sub-sequence, ping-pong by pass
    uint64_t* assumed;  // [nwg_cap] entry state each workgroup last decoded from (X_NONE: known, first of a segment)
    int4* cnt;          // [nsub_cap] (blocks started, dc sums) of the run that produced X
    int4* wsum;         // [nwg_cap] per-workgroup totals of cnt
    uint32_t* nrec;     // [nsub_cap] records (non-zero AC coefficients) of the run that produced X
    uint32_t* wrec;     // [nwg_cap] per-workgroup totals of nrec
    uint32_t* tile_start;  // compact coefficient stream (non-null): [ntiles + 1] first record of every K4 tile, preset to 0 here
    uint32_t ntiles;
    uint4* coef16;      // pass 0 clears the coefficient buffer, a slice per workgroup, behind its decode (dense layout only)
    uint64_t coef_n16;
    uint32_t* ebound;   // ... and presets K4's per-block bounds to +inf (a block K2 leaves out takes K4's exact path)
    uint32_t nblocks;
    unsigned long long* bslot;  // [nwg_cap] K2's exchange slots for blocks split over two workgroups: cleared here
    uint32_t* done;     // [nwg_cap] chained pass: workgroup g has published its final exit state
    int chained;        // this pass waits for the predecessor workgroup instead of trusting the previous pass
    uint32_t warm;      // warm-up sub-sequences (<= WARM)
    unsigned long long* part;   // K0's look-back words (cleared by the last launch)
    uint32_t nparts;
    uint32_t nwg_cap;
    int pass;
    uint32_t* status;   // [1]: error flags (a chained wait that timed out); KPEG_SYNC_STATS builds: words 8..13 collect loop counts
    unsigned long long spin_ticks;   // bound of the chained pass's wait for the predecessor (SpinGuard)
    uint32_t fault;     // test hook: bit 1 = workgroup 0 of a rippling chained pass never publishes; bit 3 = every third workgroup of k_sync_write gives the call up
    uint32_t gray;      // one-component stream (extension): see run_count; the chroma blocks' bounds are preset to "exact"
    const uint8_t* scan;   // non-null: no K0 ran -- the sub-sequences are chunks of the byte-stuffed scan (stage_unstuff), nsub = nsub_host
    uint32_t scan_len;
    uint32_t nsub_host;
    // k_sync_write (K1's pass 0 and K2 in one kernel): the call's number (never 0; 0 = not that path), and two words per
    // workgroup that take it: its presets are done / its totals and states are published
    uint32_t gen;
    uint32_t gen2;      // k_sync_write's second launch: the number its own records carry (gen: the call's, which fused_fail must carry for it to run)
    unsigned long long* pub;   // [nwg_cap][PUB_WORDS] then [nwg_cap][PUB2_WORDS]: value | call number << 32, relaxed atomics both ways
};
constexpr uint64_t X_NONE = ~0ull;
constexpr uint32_t PUB_WORDS = 9;   // blocks, dc0, dc1, dc2, records, exit state lo / hi, assumed entry state lo / hi
constexpr uint32_t PUB3_WORDS = 7;  // k_sync_write's second (strict) launch: final exit state lo / hi, then blocks, dc0, dc1, dc2, records up to and including the workgroup; behind the second records
constexpr uint32_t PUB2_WORDS = 5;  // a workgroup that repaired itself: blocks, dc0, dc1, dc2, records once more, at pub + nwg_cap PUB_WORDS

// Appends v to list[] for every lane that wants to; call with the whole wavefront converged.
__device__ __forceinline__ void push_item(bool want, uint32_t v, uint16_t* list, uint32_t* counter)
{
    const uint64_t m = __ballot(want);
    if (m == 0) return;
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    if (want) list[base + rank] = (uint16_t)v;
}

// One workgroup total (blocks, DC sums; records) for the scan below.  fresh: it may have been written in this launch
// (read past this CU's L1); else it comes from an earlier launch and one 16-byte load does.
template <bool COUNT>
__device__ __forceinline__ void wsum_load(const int4* wsum, const uint32_t* wrec, uint32_t i, uint32_t nw, bool fresh, int4& v, uint32_t& vr)
{
    v = make_int4(0, 0, 0, 0);
    vr = 0;
    if (i >= nw) return;
    if (fresh) {
        int* w = reinterpret_cast<int*>(const_cast<int4*>(wsum) + i);
        v = make_int4(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(w + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (COUNT) vr = __hip_atomic_load(const_cast<uint32_t*>(wrec) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        v = wsum[i];
        if (COUNT) vr = wrec[i];
    }
}

// Scan of (nb, dc0, dc1, dc2): workgroup totals -> exclusive prefix, by one workgroup of K1's last
// launch (SYNC_WG threads; s = 2 * SYNC_WG / 64 int4 of LDS); also the call's bookkeeping: blocks found, passes
// used, K0's look-back words cleared for the next call.  (v, vr): this thread's total of the first chunk (wsum_load of
// workgroup threadIdx.x), so that the caller can have it on its way while it does something else; every further chunk is
// loaded one chunk ahead: the workgroup that scans works alone, and what it waits for is memory.
template <int S, bool COUNT>
__device__ void wsum_scan(int4* wsum, uint32_t* wrec, EntropyMeta* meta, uint32_t nsub, uint32_t* status, int pass, bool rippling, unsigned long long* part,
                          uint32_t nparts, int4* s, int4* carry, int4 v, uint32_t vr)
{
    KPEG_GEOMETRY(S);
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (uint32_t i = t; i < nparts; i += SYNC_WG) part[i] = 0ull;
    const uint32_t nw = (nsub + OWN - 1) / OWN;
    // s[0 .. SYNC_WG / 64): the wavefronts' totals; s[SYNC_WG / 64 ..): the same for the record counts (their .x)
    uint32_t* rc = reinterpret_cast<uint32_t*>(carry) + 1;   // (carry->x is read below: a word beside it)
    int4 run = make_int4(0, 0, 0, 0);     // totals of the chunks before this one (every thread keeps its own copy)
    uint32_t runr = 0;
    for (uint32_t base = 0; base < nw; base += SYNC_WG) {
        const uint32_t i = base + t;
        int4 vn;
        uint32_t vrn;
        wsum_load<COUNT>(wsum, wrec, i + SYNC_WG, nw, rippling, vn, vrn);
        // inclusive scan inside the wavefront (DPP), the wavefronts' totals through LDS
        const int4 inc = make_int4(wave_scan_incl(v.x), wave_scan_incl(v.y), wave_scan_incl(v.z), wave_scan_incl(v.w));
        const uint32_t incr = COUNT ? wave_scan_incl(vr) : 0u;
        __syncthreads();   // (the totals of the chunk before have been read)
        if (lane == 63) {
            s[wave] = inc;
            if (COUNT) s[SYNC_WG / 64 + wave].x = (int)incr;
        }
        __syncthreads();
        int4 b = run;
        uint32_t br = runr;
        int4 tot = run;
        uint32_t totr = runr;
        for (uint32_t q = 0; q < SYNC_WG / 64; ++q) {
            const int4 wq = s[q];
            const uint32_t rq = COUNT ? (uint32_t)s[SYNC_WG / 64 + q].x : 0u;
            if (q < wave) {
                b = add4(b, wq);
                br += rq;
            }
            tot = add4(tot, wq);
            totr += rq;
        }
        if (i < nw) {
            wsum[i] = make_int4(b.x + inc.x - v.x, b.y + inc.y - v.y, b.z + inc.z - v.z, b.w + inc.w - v.w);   // exclusive
            if (COUNT) wrec[i] = br + incr - vr;
        }
        run = tot;
        runr = totr;
        v = vn;
        vr = vrn;
    }
    if (t == 0) {
        *carry = run;
        *rc = runr;
    }
    __syncthreads();
    if (t == 0) {
        meta->total_rec = *rc;
        meta->total_blocks = (uint32_t)carry->x;
        uint32_t passes = 1;   // launches of K1 that had work
        for (int q = 1; q < pass; ++q)
            if (q == 1 || meta->moved[q - 1]) passes = q + 1;
        status[2] = rippling ? (uint32_t)pass + 1 : passes;
    }
}

// Items of a workgroup: [0, wu) the last wu sub-sequences of its predecessor (warm-up, pass 0 only),
// [wu, nit) its own, one per thread.  The entry state of item j is the exit state of item j - 1.
// Pass 0: every item decodes from a guessed boundary (its own first bit, DC of component 0; the
//   first sub-sequence of a restart segment from its known state), then every wavefront re-decodes exactly the
//   items whose predecessor's exit state moved, until none moves.  Item 0's
//   guess cannot be checked here: the warm-up distance makes it irrelevant for the own items unless
//   the stream needs more than WARM_BITS to re-synchronise.
// Pass p >= 1: a workgroup whose assumed entry state differs from its predecessor's real exit
//   state re-decodes from that state and the change ripples on.
// (6 waves per SIMD = 3 workgroups per CU: all of an 8K image's workgroups run at once; the scan code at the kernel's end
// must not be allowed to raise the register count past that)
template <int S, bool COUNT, bool S420 = false>
__global__ __launch_bounds__(SYNC_WG) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_sync_pass(SyncArgs a)
{
    KPEG_GEOMETRY(S);
    __shared__ __attribute__((aligned(16))) LdsTables T;
    __shared__ int4 s_cnt[2 * (SYNC_WG / 64)];   // wsum_scan's scratch
    __shared__ __attribute__((aligned(16))) uint32_t s_lutx[S420 ? 4 : 6 * (1 << LUT_BITS)];   // run_exit's tables
    __shared__ uint64_t s_wexit[SYNC_WG / 64];   // every wavefront's last exit state so far ...
    __shared__ uint32_t s_wdone[SYNC_WG / 64];   // ... and whether it is final
    __shared__ uint64_t s_edge[2];           // entry state of the first own item, exit state of the last
    __shared__ uint32_t s_start[SYNC_WG + StuffedGeom<S>::EXTRA + 2];   // stuffed mode: first bit of every staged chunk in s_bits
    __shared__ uint32_t s_n[1];
    __shared__ int4 s_red[SYNC_WG / 64];
    __shared__ uint32_t s_redn[SYNC_WG / 64];
    constexpr uint32_t STAGE_CAP = ITEMS * SUBSEQ_WORDS + 1 + STAGE_MARGIN;
    __shared__ uint32_t s_bits[STAGE_CAP];
    const int p = a.pass;
    uint32_t g = blockIdx.x;
    const uint32_t t = threadIdx.x;
    const bool stuffed = a.scan != nullptr;
    const uint32_t nsub = stuffed ? a.nsub_host : a.meta->nsub, nseg = stuffed ? 1u : a.meta->nseg, n_u = stuffed ? 0u : a.meta->n_u;
    // The last launch (chained) also scans the workgroup totals: at once by workgroup 0 if the pass
    // before it moved nothing (the usual case), else by the workgroup that finishes the ripple last.
    if (a.gen && p >= 1) {
        if (a.meta->fused_fail != a.gen) return;   // k_sync_write finished the call
        if (p == 1 && g * OWN < nsub) {
            // what pass 0 does behind its first decode, k_sync_write left undone: the presets (it has every entry written by the
            // one workgroup that owns it, or not at all), and the exchange slots its workgroups may have used
            const uint32_t nwg = (nsub + OWN - 1) / OWN;
            const uint32_t tper = (a.ntiles + 1 + nwg - 1) / nwg;
            const uint32_t t0 = min(a.ntiles + 1, g * tper), t1 = min(a.ntiles + 1, t0 + tper);
            for (uint32_t q = t0 + t; q < t1; q += SYNC_WG) a.tile_start[q] = 0u;
            const uint32_t eper = (a.nblocks + nwg - 1) / nwg;
            const uint32_t e0 = min(a.nblocks, g * eper), e1 = min(a.nblocks, e0 + eper);
            for (uint32_t q = e0 + t; q < e1; q += SYNC_WG) a.ebound[q] = 0x7F800000u;
            if (t == 0) a.bslot[g] = 0ull;
        }
    }
    int4 pre_v = make_int4(0, 0, 0, 0);
    uint32_t pre_r = 0;
    if (a.chained && g == 0 && stuffed) wsum_load<COUNT>(a.wsum, a.wrec, t, (nsub + OWN - 1) / OWN, false, pre_v, pre_r);   // (on its way with the flag below: the usual case needs it)
    const bool rippling = p >= 2 && a.meta->moved[p - 1] != 0;
    if (a.chained && !rippling) {
        if (g == 0) {
            if (!stuffed) wsum_load<COUNT>(a.wsum, a.wrec, t, (nsub + OWN - 1) / OWN, false, pre_v, pre_r);
            wsum_scan<S, COUNT>(a.wsum, a.wrec, a.meta, nsub, a.status, p, false, a.part, a.nparts, s_cnt, &s_red[0], pre_v, pre_r);
        }
        return;
    }
    if (a.chained) {
        // a rippling chained pass waits for its predecessor: logical index by ticket (see SpinGuard)
        if (t == 0) s_n[0] = atomicAdd(&a.meta->k1_order, 1u);
        __syncthreads();
        g = s_n[0];
        __syncthreads();
    }
    const uint32_t i0 = g * OWN;
    if (i0 >= nsub) return;
    if (p >= 2 && !rippling) return;  // converged
    auto finish_chained = [&]() {
        // whole workgroup: count this workgroup as done; the last one scans
        if (t == 0) {
            __threadfence();
            s_n[0] = atomicAdd(&a.meta->ticket, 1u) == (nsub + OWN - 1) / OWN - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (s_n[0]) {
            int4 v;
            uint32_t vr;
            wsum_load<COUNT>(a.wsum, a.wrec, t, (nsub + OWN - 1) / OWN, true, v, vr);
            wsum_scan<S, COUNT>(a.wsum, a.wrec, a.meta, nsub, a.status, p, true, a.part, a.nparts, s_cnt, &s_red[0], v, vr);
        }
    };
    const uint64_t* Xb_prev = a.Xb + (size_t)((p & 1) ^ 1) * a.nwg_cap;
    uint64_t* Xb_cur = a.Xb + (size_t)(p & 1) * a.nwg_cap;
    // Chained pass (the last one enqueued, only if the pass before it still moved something): every
    // workgroup waits until its predecessor has published its final exit state, so one launch ends the
    // ripple however far it has to run (g is the order in which the workgroups started: the predecessor is
    // running or done).  Streams that re-synchronise slowly (dense noise) end here; it costs a chain
    // of workgroup decodes, but no stream is given up.
    const bool mute = a.chained && (a.fault & 2u) && g == 0;   // test hook: the successors time out
    uint64_t entry = 0;
    if (p >= 1) {
        if (a.chained && g > 0) {
            if (t == 0) {
                SpinGuard guard(a.spin_ticks);
                while (__hip_atomic_load(&a.done[g - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    if (guard.expired()) {
                        atomicOr(&a.status[1], KPEG_ERR_TIMEOUT);   // go on from whatever the predecessor has published so far
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
            entry = __hip_atomic_load(&Xb_cur[g - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (g > 0) {
            entry = Xb_prev[g - 1];
        }
        const uint64_t as = a.assumed[g];
        if (g == 0 || as == X_NONE || as == entry) {
            // the state this workgroup decoded from stands, and so do its results
            if (t == 0) {
                Xb_cur[g] = Xb_prev[g];
                if (a.chained && !mute) {
                    __threadfence();
                    __hip_atomic_store(&a.done[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (a.chained) finish_chained();
            return;
        }
    }
#define K1_BODY_BEFORE_ROUNDS
#include "k1_wg_body.inc.h"
#undef K1_BODY_BEFORE_ROUNDS
    if (a.chained) {
        __syncthreads();
        finish_chained();
    }
}

// ------------------------------------------------------------------------------------------
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
    int16_t* coef;       // cleared by K1's pass 0
    float* ebound;       // [block] K4's per-block error bound
    uint32_t nsub_cap;
    uint32_t nmcu;
    uint32_t interval;   // 0 = none
    unsigned long long* bslot;  // [nwg_cap] exchange slots, slot g: the block split between workgroups g and g + 1
    uint32_t* status;
    // compact coefficient stream (k_write<S, true>) instead of the dense layout: one 32-bit record per non-zero AC
    // coefficient in stream order -- [31:16] value, [13:8] natural position, [4:0] block within its K4 tile (24 blocks =
    // 8 MCUs) -- every block's DC in a dense int16 array, and the first record of every tile
    const uint32_t* nrec;    // [nsub_cap] K1's record count per sub-sequence
    const uint32_t* wrec;    // [nwg_cap] exclusive prefix of the per-workgroup totals
    uint32_t* rec;
    uint32_t rec_cap;
    int16_t* dc16;           // [blocks]
    uint32_t* tile_start;    // [ntiles + 1], preset to 0 by K1
    uint32_t ntiles;
    uint32_t gray;           // one-component stream (extension, dense layout only): every block is block 0 of its MCU
    const uint8_t* scan;     // non-null: no K0 ran -- the sub-sequences are chunks of the byte-stuffed scan (stage_unstuff), nsub = nsub_host
    uint32_t scan_len;
    uint32_t nsub_host;
    EntropyMeta* meta_reset; // the call's last entropy kernel leaves K1's bookkeeping zero for the next call
    uint32_t gen;            // != 0: k_sync_write ran before this launch; nothing to do unless it gave up (meta->fused_fail == gen)
};
constexpr uint32_t TILE_BLOCKS = 24;   // K4's tile: 8 MCUs x 3 components

// One lane per sub-sequence, one symbol per iteration (a flat state machine: the lanes of a
// wavefront sit at different points of different blocks).  A lane handles exactly the symbols of
// its own sub-sequence -- no lane waits for another one's long block -- and stores each
// coefficient straight into the cleared buffer (fire-and-forget 2-byte stores: no per-lane block
// buffer in LDS, so the workgroups fit many to a CU and hide each other's latency).  The entry
// state says where inside which block the lane starts (block index and DC predictors from the
// scan, coefficient position k, quirk-Q1 flag q from the exit state of its predecessor).
// K4's per-block error bound needs sums over the whole block: a block that starts and ends in one
// lane is settled there; for a block split over lanes every lane leaves its share in LDS and the
// lane that holds the block's end adds the shares up, walking back to the lane that started it.
// A block split over two workgroups: both sides swap their sum into an exchange slot, and the side that
// finds the other's sum there settles the bound.  Bounds are preset to +inf (K4's exact path), so a
// block nobody settles (corrupt stream) is still decoded correctly.
template <int S, bool COMPACT, bool S420 = false>
__global__ __launch_bounds__(SYNC_WG) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_write(WriteArgs a)
{
    KPEG_GEOMETRY(S);
    __shared__ __attribute__((aligned(16))) LdsTables T;
    __shared__ int4 s_pre[SYNC_WG];   // first the scan of cnt, then every lane's share of the block open at its exit
    __shared__ uint32_t s_prer[COMPACT ? SYNC_WG : 1];   // scan of the record counts
    __shared__ int4 s_wred[SYNC_WG / 64];
    __shared__ uint32_t s_wredr[COMPACT ? SYNC_WG / 64 : 1];
    __shared__ uint32_t s_wredr_st[SYNC_WG / 64 + 1];   // stage_unstuff's scratch
    constexpr uint32_t STAGE_CAP = SYNC_WG * SUBSEQ_WORDS + 1 + STAGE_MARGIN;
    __shared__ uint32_t s_bits[STAGE_CAP];
    __shared__ uint32_t s_start[SYNC_WG + StuffedGeom<S>::EXTRA + 2];   // stuffed mode: first bit of every staged chunk in s_bits
    const bool stuffed = a.scan != nullptr;
    const uint32_t nsub = stuffed ? a.nsub_host : a.meta->nsub, nseg = stuffed ? 1u : a.meta->nseg, n_u = stuffed ? 0u : a.meta->n_u;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.meta_reset) {
        // K1 is over (this kernel follows its last launch): its per-call bookkeeping back to zero -- K0 does this when it runs,
        // nothing else would for a call without K0
        for (int q = 0; q < SYNC_PASSES + 8; ++q) a.meta_reset->moved[q] = 0;
        a.meta_reset->ticket = 0;
        a.meta_reset->k1_order = 0;
    }
    if (a.gen && a.meta->fused_fail != a.gen) return;   // k_sync_write finished the call
    const uint32_t i0 = blockIdx.x * OWN;
    if (i0 >= nsub) return;
#if KPEG_SYNC_STATS
    const uint64_t tw0 = __builtin_amdgcn_s_memtime();
    uint32_t st_steps = 0;
#endif
    const uint32_t i = i0 + threadIdx.x;
    const bool valid = threadIdx.x < OWN && i < nsub;   // K1's partition: OWN sub-sequences per workgroup
    // everything this lane needs from K1, asked for before the tables and the slice are (one memory latency, not three)
    const int4 cnt_i = valid ? a.cnt[i] : make_int4(0, 0, 0, 0);
    const uint32_t nrec_i = COMPACT && valid ? a.nrec[i] : 0u;
    const uint64_t x_prev = valid && i > 0 ? a.X[i - 1] : 0ull;
    const int4 wsum_g = a.wsum[blockIdx.x];
    const uint32_t wrec_g = COMPACT ? a.wrec[blockIdx.x] : 0u;
    StuffedStage<S> stg;
    if (stuffed) stg.begin(a.scan, a.scan_len, i0, min((uint32_t)OWN, nsub - i0) + StuffedGeom<S>::EXTRA);   // (its loads fly while the tables load)
    load_tables(&T, a.tabs);
    SubGeom g0;
    uint32_t w0 = 0;
    if (stuffed) {
        stg.finish(s_bits, STAGE_CAP, s_start, s_wredr_st, a.scan_len);
        g0.seg = 0;
        g0.li = i0;
        g0.pstart = 0;
        g0.pend = 0;
    } else {
        g0 = sub_geom<S>(a.seg_off, a.sub_base, nseg, n_u, i0);
        w0 = g0.pstart >> 5;
        const uint32_t total_words = (n_u + 3) / 4 + 2;
        stage_bits(s_bits, STAGE_CAP, a.u, w0, total_words > w0 ? total_words - w0 : 0u);
    }
#if KPEG_SYNC_STATS
    __syncthreads();
    const uint64_t tw1 = __builtin_amdgcn_s_memtime();
#endif
#define K2_S_START s_start
#define K2_WG blockIdx.x
#include "k2_scan.inc.h"
#include "k2_core.inc.h"
#undef K2_WG
#undef K2_S_START
}

// K1's pass 0 and K2 in ONE kernel, for the case the headline is: one image without restart markers (no K0), the compact
// coefficient stream, three components, at most FUSED_MAX_WG workgroups.  A workgroup does K1's work on its sub-sequences
// (k1_wg_body.inc.h, as pass 0), publishes its totals, its last exit state and the entry state it assumed, checks that
// assumption against the exit state of the workgroup before it (and repairs itself if it was wrong: K1's work once more as a
// later pass does it, totals published a second time), then waits until every workgroup before it has published, adds their
// totals up -- that is the scan -- and writes its coefficients straight away (k2_core.inc.h), from the tables and the bits it
// has in LDS already: no second prologue, no verifying, chained or scan launch, and the workgroups in front of the image's
// slowest chain of re-decodes write while that chain is still being walked.  If a repair moves the workgroup's exit state or
// a wait expires, the workgroup writes nothing and says so in meta->fused_fail: the kernel's second launch (STRICT, see
// below) then does the call again the sure way.  Every wait is bounded (SpinGuard) and only ever for workgroups with a
// smaller index.
template <int S, bool STRICT>
__global__ __launch_bounds__(SYNC_WG) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_sync_write(SyncArgs ka, WriteArgs a)
{
    KPEG_GEOMETRY(S);
    constexpr bool COUNT = true, COMPACT = true, S420 = false;
    __shared__ __attribute__((aligned(16))) LdsTables T;
    __shared__ int4 s_cnt[2 * (SYNC_WG / 64)];
    __shared__ __attribute__((aligned(16))) uint32_t s_lutx[6 * (1 << LUT_BITS)];   // K1: run_exit's tables; K2: s_pre and s_prer
    __shared__ uint64_t s_wexit[SYNC_WG / 64];
    __shared__ uint32_t s_wdone[SYNC_WG / 64];
    __shared__ uint64_t s_edge[2];
    __shared__ uint32_t s_start[SYNC_WG + StuffedGeom<S>::EXTRA + 2];
    __shared__ uint32_t s_n[1];
    __shared__ int4 s_red[SYNC_WG / 64];
    __shared__ uint32_t s_redn[SYNC_WG / 64];
    constexpr uint32_t STAGE_CAP = ITEMS * SUBSEQ_WORDS + 1 + STAGE_MARGIN;
    __shared__ uint32_t s_bits[STAGE_CAP];
    static_assert(sizeof(s_lutx) >= SYNC_WG * (sizeof(int4) + sizeof(uint32_t)), "K2's scan arrays take the place of run_exit's tables");
    (void)s_cnt;
    // the hand-over from the workgroups before: every wavefront's part of their totals, its last record's exit state (two rounds), x_before
    __shared__ int4 s_hsum[SYNC_WG / 64];
    __shared__ uint32_t s_hrec[SYNC_WG / 64];
    __shared__ uint64_t s_hx[2 * (SYNC_WG / 64) + 1];
    const bool stuffed = true;   // (the host takes this path for such calls only)
    const uint32_t nsub = ka.nsub_host, nseg = 1u, n_u = 0u;
    // The kernel is enqueued twice.  The second launch (STRICT) leaves at once unless the first gave the call up (a stream that does not
    // re-synchronise inside a workgroup's sub-sequences, an expired wait): then it does the call again the slow, sure way -- every workgroup
    // waits for the FINAL exit state and the totals up to its predecessor before it settles its own sub-sequences from the results the first
    // launch left, one after the other along the stream (the bits and tables are staged before the wait: a link of that chain is a look
    // and a publication, a microsecond, unless something has to be decoded again), then writes.  Its workgroups take their index from a
    // ticket, so a predecessor is always running or done whatever the dispatch order.  (Rounds 1-3 had three launches here -- verifying,
    // chained, k_write -- which cost the headline 6 us of doing nothing: profiles/r03_i.)
    uint32_t gi_ = blockIdx.x;
    if constexpr (STRICT) {
        if (ka.meta->fused_fail != ka.gen) return;
        if (threadIdx.x == 0) s_n[0] = atomicAdd(&ka.meta->strict_order, 1u);
        __syncthreads();
        gi_ = s_n[0];
        __syncthreads();
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        ka.meta->strict_order = 0u;
    }
    const uint32_t gi = gi_, ti = threadIdx.x;
    const uint32_t i0 = gi * OWN;
    if (i0 >= nsub) return;
    if constexpr (STRICT) {
        if (ti == 0) atomicExch(&ka.bslot[gi], 0ull);   // (the exchange slot with the next workgroup: the first launch may have used it)
    }
#ifndef KPEG_FUSED_PRIO
#define KPEG_FUSED_PRIO 2
#endif
    // K1's part is what the kernel waits for -- its chains of re-decodes, one lane after the other; K2's loops are bulk work that fills
    // what issue slots are left.  On a SIMD the oldest wavefront is served first, so the workgroups dispatched last (they share their CU
    // with two older ones, which are at their write loops by then) had their K1 parts end 25 us after the first ones' (tools/fused_timeline.py).
    // Priority outranks age: K1 runs raised, K2 does not.
    __builtin_amdgcn_s_setprio(KPEG_FUSED_PRIO);
    // ---- K1, pass 0 ----
    uint64_t k1_exit = 0, k1_as = 0;
    int4 k1_cnt = make_int4(0, 0, 0, 0);
    uint32_t k1_nrec = 0, k1_wu = 0;
    bool k1_own = false;
    uint64_t x_before = 0;   // the exit state of the sub-sequence before this workgroup's first
    if constexpr (STRICT) {
        const SyncArgs& a = ka;
        const int p = 1;
        const uint32_t g = gi, t = threadIdx.x;
        uint64_t* const Xb_cur = a.Xb;
        const uint64_t* const Xb_prev = a.Xb;   // (what a later pass compares and counts for the launch behind it: nothing here)
        const bool mute = false;
        uint64_t entry = 0;
        auto strict_wait = [&]() {
            if (t == 0) {
                unsigned long long v[PUB3_WORDS];
#pragma unroll
                for (uint32_t q = 0; q < PUB3_WORDS; ++q) v[q] = 0;
                if (g > 0) {
                    const unsigned long long* pw = a.pub + (size_t)a.nwg_cap * (PUB_WORDS + PUB2_WORDS) + (size_t)(g - 1) * PUB3_WORDS;
                    SpinGuard guard(K1_SPIN_TICKS);
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (uint32_t q = 0; q < PUB3_WORDS; ++q) {
                            v[q] = __hip_atomic_load(pw + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = ok && (uint32_t)(v[q] >> 32) == a.gen2;
                        }
                        if (ok) break;
                        if (guard.expired()) {   // (cannot happen: the predecessor by ticket is running or done; bounded like every wait)
                            atomicOr(&a.status[1], KPEG_ERR_TIMEOUT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
                s_hx[2 * (SYNC_WG / 64)] = (uint64_t)(uint32_t)v[0] | ((uint64_t)(uint32_t)v[1] << 32);
                s_hsum[0] = make_int4((int)(uint32_t)v[2], (int)(uint32_t)v[3], (int)(uint32_t)v[4], (int)(uint32_t)v[5]);
                s_hrec[0] = (uint32_t)v[6];
            }
            __syncthreads();
            entry = s_hx[2 * (SYNC_WG / 64)];
        };
#define K1_BODY_BEFORE_ROUNDS strict_wait();
#include "k1_wg_body.inc.h"
#undef K1_BODY_BEFORE_ROUNDS
        k1_exit = r.exit_state;
        k1_cnt = r.cnt;
        k1_nrec = r.nrec;
        k1_wu = wu;
        k1_own = have && t >= wu;
        x_before = entry;
        __syncthreads();
        if (t == 0) {
            // final: this workgroup's last exit state and the totals up to and including it
            const int4 w = a.wsum[g];   // (this thread's own stores)
            const uint32_t wr = a.wrec[g];
            const int4 pre = s_hsum[0];
            const uint64_t last = s_edge[1];
            const uint32_t vals[PUB3_WORDS] = {(uint32_t)last, (uint32_t)(last >> 32), (uint32_t)(pre.x + w.x), (uint32_t)(pre.y + w.y), (uint32_t)(pre.z + w.z), (uint32_t)(pre.w + w.w), s_hrec[0] + wr};
            unsigned long long* const p3 = a.pub + (size_t)a.nwg_cap * (PUB_WORDS + PUB2_WORDS) + (size_t)g * PUB3_WORDS;
            for (uint32_t q = 0; q < PUB3_WORDS; ++q)
                __hip_atomic_store(p3 + q, (unsigned long long)vals[q] | ((unsigned long long)a.gen2 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        const SyncArgs& a = ka;
        const int p = 0;
        const uint32_t g = gi, t = threadIdx.x;
        uint64_t* const Xb_cur = a.Xb;
        const uint64_t* const Xb_prev = a.Xb + a.nwg_cap;
        const bool mute = false;
        const uint64_t entry = 0;
        (void)Xb_prev;
#define K1_BODY_BEFORE_ROUNDS
#include "k1_wg_body.inc.h"
#undef K1_BODY_BEFORE_ROUNDS
        k1_exit = r.exit_state;
        k1_cnt = r.cnt;
        k1_nrec = r.nrec;
        k1_wu = wu;
        k1_own = have && t >= wu;
        k1_as = s_edge[0];   // (what this workgroup's first own item decoded from: written before the body's last barrier)
        // (the body's thread 0 has published the totals, the last exit state and the assumption: a.gen != 0)
    }
    const uint64_t k1_last = s_edge[1];   // this workgroup's last exit state, as published (written before the body's last barrier)
    // The exit state this workgroup's first own item has to have started from: the last one of the workgroup before.  Asked for now,
    // looked at behind the hand-over below (thread 0).
    unsigned long long xq0 = 0, xq1 = 0;
    if (!STRICT && ti == 0 && gi > 0) {
        xq0 = __hip_atomic_load(ka.pub + (size_t)(gi - 1) * PUB_WORDS + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xq1 = __hip_atomic_load(ka.pub + (size_t)(gi - 1) * PUB_WORDS + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#if KPEG_SYNC_STATS
    const uint64_t tw0 = __builtin_amdgcn_s_memtime();   // K1's part is over for the whole workgroup, its totals are published
#endif
    // ---- what K1's lanes know, handed to K2's lanes: item j of the workgroup sits on lane j + wu there, on lane j here; then K2's
    // scan inside the workgroup.  None of it needs another workgroup: it is done before the waits below, not behind them. ----
    int4* const s_pre = reinterpret_cast<int4*>(s_lutx);
    uint32_t* const s_prer = s_lutx + SYNC_WG * 4;
    int4* const s_wred = s_red;
    uint32_t* const s_wredr = s_redn;
    uint64_t* const s_x = reinterpret_cast<uint64_t*>(s_lutx);
    const uint32_t i = i0 + ti;
    const bool valid = ti < OWN && i < nsub;
    uint64_t x_prev = 0;
    int4 cnt_i = make_int4(0, 0, 0, 0);
    uint32_t nrec_i = 0;
    auto hand_over = [&]() {
        __syncthreads();   // (K1's last readers of s_red / s_redn and of s_lutx are done)
        if (k1_own) s_x[ti - k1_wu] = k1_exit;
        __syncthreads();
        x_prev = valid && ti > 0 ? s_x[ti - 1] : 0ull;   // (lane 0's is the predecessor workgroup's last exit state: below)
        __syncthreads();
        if (k1_own) {
            s_pre[ti - k1_wu] = k1_cnt;
            s_prer[ti - k1_wu] = k1_nrec;
        }
        __syncthreads();
        cnt_i = valid ? s_pre[ti] : make_int4(0, 0, 0, 0);
        nrec_i = valid ? s_prer[ti] : 0u;
        __syncthreads();
#include "k2_scan.inc.h"
    };
    hand_over();
    __builtin_amdgcn_s_setprio(0);
    int f_bad = 0;
#if KPEG_SYNC_STATS
    uint64_t tw1 = 0;
#endif
    if constexpr (!STRICT) {
    // ---- this workgroup's own assumption ----
    // Its first own item decoded from the state its lead-in items arrived at; the state it had to start from is the last exit state of the
    // workgroup before (final as published: a workgroup's last item has re-synchronised hundreds of items after its first, whatever that
    // one started from).  Synthetic fields always match.  Photographs have, somewhere, a workgroup whose lead-in had not re-synchronised
    // yet: that workgroup REPAIRS itself here -- K1's work once more as a later pass does it (the results loaded, the first item's entry
    // state set right, the items it reaches decoded again) -- and publishes its totals a second time, in a record of their own.  (Round 2
    // gave the whole call up to the three launches behind this kernel: 0.30 instead of 0.21 ms for an 8K photograph at 1.5 bit/px.)
    if (ti == 0 && gi > 0) {
        SpinGuard guard(S >= SUBSEQ_DENSE && SUBSEQ_DENSE > SUBSEQ_SPARSE ? FUSED_SPIN_TICKS_DENSE : FUSED_SPIN_TICKS);
        while ((uint32_t)(xq0 >> 32) != ka.gen || (uint32_t)(xq1 >> 32) != ka.gen) {
            if (guard.expired()) {
                f_bad = 2;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
            xq0 = __hip_atomic_load(ka.pub + (size_t)(gi - 1) * PUB_WORDS + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xq1 = __hip_atomic_load(ka.pub + (size_t)(gi - 1) * PUB_WORDS + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_hx[2 * (SYNC_WG / 64)] = (uint64_t)(uint32_t)xq0 | ((uint64_t)(uint32_t)xq1 << 32);   // x_before
    }
    if (__syncthreads_or(f_bad)) {
        if (ti == 0) __hip_atomic_store(&ka.meta->fused_fail, ka.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    x_before = gi > 0 ? s_hx[2 * (SYNC_WG / 64)] : 0ull;
    if (gi > 0 && k1_as != X_NONE && k1_as != x_before) {   // (the same for every thread)
        __builtin_amdgcn_s_setprio(KPEG_FUSED_PRIO);
        {
            const SyncArgs& a = ka;
            const int p = 1;
            const uint32_t g = blockIdx.x, t = threadIdx.x;
            uint64_t* const Xb_cur = a.Xb;
            const uint64_t* const Xb_prev = a.Xb;   // (what a later pass compares and counts for the launch behind it: nothing here)
            const bool mute = false;
            const uint64_t entry = x_before;
#define K1_BODY_BEFORE_ROUNDS
#include "k1_wg_body.inc.h"
#undef K1_BODY_BEFORE_ROUNDS
            k1_exit = r.exit_state;
            k1_cnt = r.cnt;
            k1_nrec = r.nrec;
            k1_wu = wu;
            k1_own = have && t >= wu;
        }
        __syncthreads();
        // the last exit state has been published and compared: it must stand (it does, unless the stream never re-synchronises inside
        // a workgroup's 501 sub-sequences: the launches behind this kernel take such a call)
        if (s_edge[1] != k1_last) {
            if (ti == 0) __hip_atomic_store(&ka.meta->fused_fail, ka.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (ti == 0) {
            const int4 w = ka.wsum[gi];   // (this thread's own stores)
            const uint32_t wr = ka.wrec[gi];
            const uint32_t vals[PUB2_WORDS] = {(uint32_t)w.x, (uint32_t)w.y, (uint32_t)w.z, (uint32_t)w.w, wr};
            unsigned long long* const p2 = ka.pub + (size_t)ka.nwg_cap * PUB_WORDS + (size_t)gi * PUB2_WORDS;
            for (uint32_t q = 0; q < PUB2_WORDS; ++q)
                __hip_atomic_store(p2 + q, (unsigned long long)vals[q] | ((unsigned long long)ka.gen << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&ka.meta->repaired, 1u);
        }
        hand_over();
        __builtin_amdgcn_s_setprio(0);
    }
    // ---- every workgroup before this one: wait for what it published, add it up ----
    // Thread ti takes workgroups ti, ti + SYNC_WG, ..., two at a time.  A record's nine words (and the two of the record before it that
    // hold the exit state its assumption has to match) are asked for together and looked at afterwards -- each says for itself whether
    // it is there (value | call number << 32) -- and a thread polls until its records are whole: when the image's slowest workgroup
    // publishes at last, the workgroups behind it have everything else in registers already and are one load away from their write
    // loops.  (Round 2 waited for first words with one wavefront and 2 us of sleep between looks, then read every word through its
    // own validation loop: eleven round trips one after the other, 25 us between the slowest K1 part's end and the K2 loops behind it.)
    // A workgroup whose assumption does not match the exit state before it (seen here as it sees it itself) repairs itself: its totals
    // are taken from its second record, when that is there.
    // The grid may be larger than what the device holds at once (photographs at 8K: 1000-2500 workgroups for 768 places).  A workgroup only
    // ever waits for smaller indices; the dispatcher hands workgroups out in index order, so those are running or done -- and if a device
    // did not, the waits expire (SpinGuard) and the launches behind this kernel decode the call: slower, never wrong, never hung.
    int4 f_sum = make_int4(0, 0, 0, 0);
    uint32_t f_rec = 0;
    {
        SpinGuard guard(S >= SUBSEQ_DENSE && SUBSEQ_DENSE > SUBSEQ_SPARSE ? FUSED_SPIN_TICKS_DENSE : FUSED_SPIN_TICKS);
        auto load_rec = [&](uint32_t h, int4& sm, uint32_t& rc, bool& rep) -> bool {
            const unsigned long long* pw = ka.pub + (size_t)h * PUB_WORDS;
            unsigned long long v[PUB_WORDS], l0 = (unsigned long long)ka.gen << 32, l1 = l0;
#pragma unroll
            for (uint32_t q = 0; q < PUB_WORDS; ++q) v[q] = __hip_atomic_load(pw + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (h > 0) {
                l0 = __hip_atomic_load(pw - PUB_WORDS + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                l1 = __hip_atomic_load(pw - PUB_WORDS + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bool ok = (uint32_t)(l0 >> 32) == ka.gen && (uint32_t)(l1 >> 32) == ka.gen;
#pragma unroll
            for (uint32_t q = 0; q < PUB_WORDS; ++q) ok = ok && (uint32_t)(v[q] >> 32) == ka.gen;
            if (ok) {
                sm = make_int4((int)(uint32_t)v[0], (int)(uint32_t)v[1], (int)(uint32_t)v[2], (int)(uint32_t)v[3]);
                rc = (uint32_t)v[4];
                const uint64_t as = (uint64_t)(uint32_t)v[7] | ((uint64_t)(uint32_t)v[8] << 32), left = (uint64_t)(uint32_t)l0 | ((uint64_t)(uint32_t)l1 << 32);
                rep = h > 0 && as != X_NONE && as != left;   // that workgroup has repaired itself (or is about to)
            }
            return ok;
        };
        auto load_rec2 = [&](uint32_t h, int4& sm, uint32_t& rc) -> bool {
            const unsigned long long* pw = ka.pub + (size_t)ka.nwg_cap * PUB_WORDS + (size_t)h * PUB2_WORDS;
            unsigned long long v[PUB2_WORDS];
#pragma unroll
            for (uint32_t q = 0; q < PUB2_WORDS; ++q) v[q] = __hip_atomic_load(pw + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (uint32_t q = 0; q < PUB2_WORDS; ++q) ok = ok && (uint32_t)(v[q] >> 32) == ka.gen;
            if (ok) {
                sm = make_int4((int)(uint32_t)v[0], (int)(uint32_t)v[1], (int)(uint32_t)v[2], (int)(uint32_t)v[3]);
                rc = (uint32_t)v[4];
            }
            return ok;
        };
        for (uint32_t h0 = ti; h0 < gi && !f_bad; h0 += 2 * SYNC_WG) {
            const bool hb = h0 + SYNC_WG < gi;
            int4 sum_a = make_int4(0, 0, 0, 0), sum_b = make_int4(0, 0, 0, 0);
            uint32_t rec_a = 0, rec_b = 0;
            bool oka = false, okb = !hb, rep_a = false, rep_b = false;
            for (;;) {
                if (!oka) oka = load_rec(h0, sum_a, rec_a, rep_a);
                if (!okb) okb = load_rec(h0 + SYNC_WG, sum_b, rec_b, rep_b);
                if (oka && okb) break;
                if (guard.expired()) {
                    f_bad = 2;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            while ((rep_a || rep_b) && !f_bad) {
                if (rep_a) rep_a = !load_rec2(h0, sum_a, rec_a);
                if (rep_b) rep_b = !load_rec2(h0 + SYNC_WG, sum_b, rec_b);
                if (!(rep_a || rep_b)) break;
                if (guard.expired() || __hip_atomic_load(&ka.meta->fused_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ka.gen) {   // (or that workgroup gave the call up)
                    f_bad = 2;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            f_sum = add4(f_sum, add4(sum_a, sum_b));
            f_rec += rec_a + rec_b;
        }
    }
#if KPEG_SYNC_STATS
    tw1 = __builtin_amdgcn_s_memtime();   // this wavefront has every record it takes
#endif
    {
        const uint32_t lane = ti & 63, wave = ti >> 6;
        f_sum = make_int4(wave_scan_incl(f_sum.x), wave_scan_incl(f_sum.y), wave_scan_incl(f_sum.z), wave_scan_incl(f_sum.w));
        f_rec = wave_scan_incl(f_rec);
        if (lane == 63) {
            s_hsum[wave] = f_sum;
            s_hrec[wave] = f_rec;
        }
    }
    if ((ka.fault & 8u) && gi % 3u == 1u) f_bad = 2;   // test hook: every third workgroup gives the call up here, its neighbours have written or will
    if (__syncthreads_or(f_bad)) {
        if (ti == 0) __hip_atomic_store(&ka.meta->fused_fail, ka.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    }
    int4 wsum_g = make_int4(0, 0, 0, 0);
    uint32_t wrec_g = 0;
    if constexpr (STRICT) {
        if (gi == 0 && ti == 0) ka.status[2] = 2u;   // (launches of K1 that had work)
        wsum_g = s_hsum[0];
        wrec_g = s_hrec[0];
    } else {
        if (gi == 0 && ti == 0) ka.status[2] = 1u;   // launches of K1 that had work (the second launch says more if it runs)
        for (uint32_t q = 0; q < SYNC_WG / 64; ++q) {
            wsum_g = add4(wsum_g, s_hsum[q]);
            wrec_g += s_hrec[q];
        }
    }
    if (valid && ti == 0 && i > 0) x_prev = x_before;
    SubGeom g0;
    g0.seg = 0;
    g0.li = i0;
    g0.pstart = 0;
    g0.pend = 0;
    const uint32_t w0 = 0;
#if KPEG_SYNC_STATS
    uint32_t st_steps = 0;
#endif
    uint32_t* const k2_start = s_start + k1_wu;
#define K2_S_START k2_start
#define K2_WG gi
#include "k2_core.inc.h"
#undef K2_WG
#undef K2_S_START
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
template <int SB>
static int entropy_decode_launch_s(EntropyScratch* S, const EntropyTables& tabs, const EntropyLaunch& L, hipEvent_t* ev, bool* ev_rec,
                                   std::string* err)
{
    KPEG_GEOMETRY(SB);
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
    const uint64_t len64 = L.nimg ? L.total_len : (uint64_t)L.scan_len;
    if (len64 >= (1ull << 28)) {
        *err = "entropy-coded data larger than 256 MiB";
        return KPEG_HIP_E_UNSUPPORTED;
    }
    const uint32_t n = (uint32_t)len64;
    const uint32_t nseg_expected = L.restart_interval ? (L.nmcu + L.restart_interval - 1) / L.restart_interval : 1;
    const uint32_t nparts = L.nimg ? L.total_parts : (n + US_BLOCK_BYTES - 1) / US_BLOCK_BYTES;
    const uint32_t nsub_cap = (uint32_t)(((uint64_t)n * 8 + SUBSEQ_BITS - 1) / SUBSEQ_BITS) + nseg_expected + 1;
    const uint32_t seg_cap = nseg_expected + 2;
    const uint32_t nwg_cap = (nsub_cap + OWN - 1) / OWN;
    int rc;
    if ((rc = ent_grow(&S->d_u, &S->u_cap, (size_t)n + 64, L.stream, err))) return rc;
    {
        void* const before = S->d_part;
        if ((rc = ent_grow(&S->d_part, &S->part_cap, (size_t)nparts * sizeof(unsigned long long), L.stream, err))) return rc;
        if (S->d_part != before) S->part_clean = false;
    }
    if ((rc = ent_grow(&S->d_segoff, &S->seg_cap, (size_t)seg_cap * 2 * sizeof(uint32_t), L.stream, err))) return rc;
    if ((rc = ent_grow(&S->d_state, &S->state_cap, ((size_t)nsub_cap + 5 * (size_t)nwg_cap) * 8 + 64, L.stream, err))) return rc;
    if ((rc = ent_grow(&S->d_cnt, &S->cnt_cap, (size_t)nsub_cap * 16, L.stream, err))) return rc;
    if ((rc = ent_grow(&S->d_wsum, &S->wsum_cap, ((size_t)nwg_cap + 2) * 16, L.stream, err))) return rc;
    if ((rc = ent_grow(&S->d_nrec, &S->nrec_cap, ((size_t)nsub_cap + (size_t)nwg_cap + 2) * 4, L.stream, err))) return rc;
    if (!S->d_meta) {
        ENT_HIP(hipMalloc((void**)&S->d_meta, sizeof(EntropyMeta)));
        ENT_HIP(hipMemset(S->d_meta, 0, sizeof(EntropyMeta)));   // K0's tickets start at zero and are left at zero
    }
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
    uint64_t* X = (uint64_t*)S->d_state;
    uint64_t* Xb = X + nsub_cap;
    uint64_t* assumed = Xb + 2 * (size_t)nwg_cap;
    unsigned long long* bslot = (unsigned long long*)(assumed + nwg_cap);
    uint32_t* done = (uint32_t*)(bslot + nwg_cap);
    int4* cnt = (int4*)S->d_cnt;
    const int rst = L.restart_interval ? 1 : 0;   // restart segments (markers in the stream, or the images of a fused batch)
    const int markers = rst && !L.nimg;            // K0 strips RSTn
    UnstuffBatch bt;
    bt.nimg = L.nimg;
    bt.scan_tab = L.d_scan_tab;
    bt.len_tab = L.d_len_tab;
    bt.wg_tab = L.d_wg_tab;

    // One image without restart markers needs no K0: the sub-sequences are chunks of the scan as it is, and the workgroups of
    // K1 and K2 un-stuff what they stage (stage_unstuff).  Restart segments and batches keep K0: it finds the markers and
    // lays the segments out.
    const bool stuffed = !L.nimg && !rst && !L.force_k0 && n > 0;
    const uint32_t nsub_host = (uint32_t)(((uint64_t)n + SUBSEQ_BITS / 8 - 1) / (SUBSEQ_BITS / 8));
    if (!stuffed) {
        if (!S->part_clean) ENT_HIP(hipMemsetAsync(S->d_part, 0, S->part_cap, L.stream));
        S->part_clean = false;
        hipLaunchKernelGGL(k_unstuff, dim3(nparts), dim3(US_THREADS), 0, L.stream, L.d_scan, n, markers, (unsigned long long*)S->d_part,
                           (uint8_t*)S->d_u, seg_off, seg_cap, S->d_meta, sub_base, nsub_cap, L.d_status, bt, nseg_expected, (uint32_t)SUBSEQ_BITS,
                           L.spin_ticks ? L.spin_ticks : K0_SPIN_TICKS, L.fault);
    }
    mark(1);

    SyncArgs sa;
    sa.u = (const uint32_t*)S->d_u;
    sa.seg_off = seg_off;
    sa.sub_base = sub_base;
    sa.meta = S->d_meta;
    sa.tabs = S->d_tabs;
    sa.X = X;
    sa.Xb = Xb;
    sa.assumed = assumed;
    sa.cnt = cnt;
    sa.wsum = (int4*)S->d_wsum;
    sa.nrec = (uint32_t*)S->d_nrec;
    sa.wrec = sa.nrec + nsub_cap;
    sa.tile_start = L.d_tile_start;
    sa.ntiles = L.ntiles;
    sa.coef16 = (uint4*)L.d_coef;
    sa.coef_n16 = (uint64_t)L.nmcu * (L.sub420 ? 48 : 24);   // 384 bytes per MCU (4:2:0: 768)
    sa.ebound = (uint32_t*)L.d_ebound;
    sa.nblocks = L.nmcu * (L.sub420 ? 6 : 3);
    sa.bslot = bslot;
    sa.done = done;
    sa.warm = L.warm < 0 ? (uint32_t)WARM : min((uint32_t)L.warm, (uint32_t)WARM);
    sa.nwg_cap = nwg_cap;
    sa.status = L.d_status;
    sa.spin_ticks = L.spin_ticks ? L.spin_ticks : K1_SPIN_TICKS;
    sa.fault = L.fault;
    sa.gray = L.gray;
    sa.scan = stuffed ? L.d_scan : nullptr;
    sa.scan_len = n;
    sa.nsub_host = nsub_host;
    sa.gen = 0;
    sa.pub = nullptr;
    WriteArgs wa;
    wa.u = (const uint32_t*)S->d_u;
    wa.seg_off = seg_off;
    wa.sub_base = sub_base;
    wa.meta = S->d_meta;
    wa.tabs = S->d_tabs;
    wa.X = X;
    wa.cnt = cnt;
    wa.wsum = (const int4*)S->d_wsum;
    wa.coef = L.d_coef;
    wa.ebound = L.d_ebound;
    wa.nsub_cap = nsub_cap;
    wa.nmcu = L.nmcu;
    wa.interval = L.restart_interval;
    wa.bslot = bslot;
    wa.status = L.d_status;
    wa.nrec = sa.nrec;
    wa.wrec = sa.wrec;
    wa.rec = L.d_rec;
    wa.rec_cap = L.rec_cap;
    wa.dc16 = L.d_dc16;
    wa.tile_start = L.d_tile_start;
    wa.ntiles = L.ntiles;
    wa.gray = L.gray;
    wa.scan = sa.scan;
    wa.scan_len = n;
    wa.nsub_host = nsub_host;
    wa.meta_reset = S->d_meta;
    wa.gen = 0;
    const int npass = L.sync_passes >= 3 ? L.sync_passes : SYNC_PASSES;   // the last one is chained and runs the scan
    sa.part = (unsigned long long*)S->d_part;
    sa.nparts = stuffed ? 0u : nparts;   // (K0's look-back words: untouched without K0)
    // One kernel for K1's pass 0 and K2 (k_sync_write) where it applies; the launches below follow it in any case and leave
    // at once unless it gave up.
    bool fuse = false, fuse_strict = false;
    // (both sub-sequence sizes since round 3: with the long ones an 8K photograph at 3-4 bit/px saves the verifying launch's second
    // decode and K2's prologue, 1-4 % of the call: profiles/r03_e)
#ifdef KPEG_NO_FUSE_DENSE
    if constexpr (SB < SUBSEQ_DENSE) {
#else
    if constexpr (true) {
#endif
        fuse = stuffed && L.d_tile_start && !L.gray && !L.sub420 && L.sync_passes == 0 && L.fused_slots && nwg_cap <= FUSED_MAX_WG;
        if (fuse) {
            const size_t cap_before = S->flags_cap;
            if ((rc = ent_grow(&S->d_flags, &S->flags_cap, (size_t)nwg_cap * (PUB_WORDS + PUB2_WORDS + PUB3_WORDS) * sizeof(unsigned long long), L.stream, err))) return rc;
            bool wipe = S->flags_cap != cap_before;   // (a new allocation -- even at the old address -- holds anything)
            if (S->gen >= 0xFFFFFFFDu) {
                S->gen = 0;   // the call numbers start over: nothing an earlier call published may pass for this one's
                wipe = true;
            }
            S->gen += 2;   // (never 0: 0 is "not that path"; the second launch's records carry the number after the call's)
            if (wipe) ENT_HIP(hipMemsetAsync(S->d_flags, 0, S->flags_cap, L.stream));
            sa.gen = wa.gen = S->gen;
            sa.gen2 = S->gen + 1;
            sa.pub = (unsigned long long*)S->d_flags;
            sa.pass = 0;
            sa.chained = 0;
            hipLaunchKernelGGL((k_sync_write<SB, false>), dim3(nwg_cap), dim3(SYNC_WG), 0, L.stream, sa, wa);
#ifndef KPEG_FUSED_THREE_LAUNCHES
            // ... and once more, the sure way, if that launch gave the call up (it leaves at once if not)
            fuse_strict = true;
            hipLaunchKernelGGL((k_sync_write<SB, true>), dim3(nwg_cap), dim3(SYNC_WG), 0, L.stream, sa, wa);
#endif
        }
    }
    for (int t = fuse ? 1 : 0; t < npass && !fuse_strict; ++t) {
        sa.pass = t;
        sa.chained = t == npass - 1 ? 1 : 0;
        if (L.sub420) hipLaunchKernelGGL((k_sync_pass<SB, false, true>), dim3(nwg_cap), dim3(SYNC_WG), 0, L.stream, sa);
        else if (L.d_tile_start) hipLaunchKernelGGL((k_sync_pass<SB, true>), dim3(nwg_cap), dim3(SYNC_WG), 0, L.stream, sa);
        else hipLaunchKernelGGL((k_sync_pass<SB, false>), dim3(nwg_cap), dim3(SYNC_WG), 0, L.stream, sa);
    }
    mark(2);
    mark(3);
    if (fuse_strict) {
        // (k_sync_write's two launches have written the coefficients)
    } else if (L.sub420) hipLaunchKernelGGL((k_write<SB, false, true>), dim3(nwg_cap), dim3(SYNC_WG), 0, L.stream, wa);
    else if (L.d_tile_start) hipLaunchKernelGGL((k_write<SB, true>), dim3(nwg_cap), dim3(SYNC_WG), 0, L.stream, wa);
    else hipLaunchKernelGGL((k_write<SB, false>), dim3(nwg_cap), dim3(SYNC_WG), 0, L.stream, wa);
    mark(4);
    mark(5);
    ENT_HIP(hipGetLastError());
    if (!stuffed) S->part_clean = true;   // K1's last launch is enqueued
    return KPEG_HIP_OK;
#undef ENT_HIP
}

// Dense streams (from 3 bits per pixel) re-synchronise over thousands of bits: K1 then needs many rounds, and
// four times longer sub-sequences mean a quarter of the rounds (each with its fixed cost) for the same
// sequential chain; K2 pays ~40-70 % for them, K1 gains more (measured 1..19 bits per pixel; the
// switch-over by K1 + K2 on synthetic fields and on tiled photographs, 4K and 8K: tools/subseq_choice.py).
// the one rule for the sub-sequence size (the launcher below and the caller's choice of the coefficient layout both go by it)
static bool entropy_dense_subseq(int forced, bool sub420, uint64_t bytes, uint64_t nmcu)
{
    const uint64_t bits = bytes * 8, px = nmcu * 64;
    // (... and from 2.25 bits per pixel where the short sub-sequences would make more than 1.5 times the workgroups the device holds at
    // once -- 6.9 MB of scan: the one kernel then runs in three and more generations, each as long as its slowest chain; 8K photographs,
    // round 3: 2.04 bit/px 0.397 / 0.400 ms short / long, 2.87 bit/px 0.547 / 0.410, 3.04 bit/px 0.611 / 0.529)
    return forced ? forced >= SUBSEQ_DENSE : (sub420 || bits >= px * 4 || (bits >= px * 3 && px >= (4u << 20)) || (bits * 4 >= px * 9 && bytes > 6900000u));
}

#ifndef KPEG_SMALL_MAX_BYTES
#define KPEG_SMALL_MAX_BYTES 2200000u   // scan bytes up to which the 64-bit sub-sequences are taken (see SUBSEQ_SMALL)
#endif
static int entropy_decode_launch(EntropyScratch* S, const EntropyTables& tabs, const EntropyLaunch& L, hipEvent_t* ev, bool* ev_rec,
                                 std::string* err)
{
    const uint64_t bytes = L.nimg ? L.total_len : (uint64_t)L.scan_len;
    // from 3 to 4 bits per pixel; 4:2:0 always: four luma blocks in a row share their tables, a decoder that is one block off
    // stays plausible until the chroma blocks come, and streams re-synchronise several times more slowly (12 Mpixel:
    // K1 1.13 ms with 384-bit sub-sequences, 1.54 ms with 96)
    // (from 3 bits per pixel on pictures of 4 Mpixel and more, where somewhere a chain of rounds is long; from 4 on small ones:
    // lena.jpg, 512 x 512 at 3.2 bits per pixel, takes 0.15 ms with the short sub-sequences and 0.19 ms with the long ones)
    const bool dense = entropy_dense_subseq(L.subseq, L.sub420 != 0, bytes, L.sub420 ? (uint64_t)L.nmcu * 4 : (uint64_t)L.nmcu);   // (a 4:2:0 MCU is 256 pixels)
    if (dense) return entropy_decode_launch_s<SUBSEQ_DENSE>(S, tabs, L, ev, ev_rec, err);
    if constexpr (SUBSEQ_SMALL < SUBSEQ_SPARSE) {
        // one small picture below 2.5 bits per pixel (a batch is one long stream, however small its pictures; restart segments keep K0)
        const uint64_t bits = bytes * 8, px = (uint64_t)L.nmcu * 64;
        const bool small = L.subseq ? L.subseq == SUBSEQ_SMALL : (!L.nimg && !L.restart_interval && bytes <= KPEG_SMALL_MAX_BYTES && bits * 2 < px * 5);
        if (small) return entropy_decode_launch_s<SUBSEQ_SMALL>(S, tabs, L, ev, ev_rec, err);
    }
    return entropy_decode_launch_s<SUBSEQ_SPARSE>(S, tabs, L, ev, ev_rec, err);
}

}  // namespace kpeg_dev
