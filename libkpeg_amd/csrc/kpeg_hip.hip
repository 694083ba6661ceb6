// libkpeg_amd/csrc/kpeg_hip.hip -- C ABI (include/kpeg_hip.h) over the gfx950 kernels.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see
// libkpeg_amd/build.py).  -ffp-contract=off is part of the contract: the exact path
// must execute the reference's float/double operations one by one.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

#include "../../include/kpeg_hip.h"
#include "entropy.hip.h"
#include "idct_colour.hip.h"
#include "kpeg_tables.h"

using namespace kpeg_dev;

static const size_t STATUS_WORDS = KPEG_STATUS_WORDS, STATUS_BYTES = STATUS_WORDS * 4;

// ---------------------------------------------------------------------------------------------
struct kpeg_hip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string last_error;
    int idct_mode = 0;
    int sync_passes = 0;  // 0 = default number of enqueued sync passes
    int warm = -1;        // test hook: K1's warm-up sub-sequences per workgroup (< 0 = default)
    int subseq = 0;       // test hook: K1/K2 sub-sequence size (0 = from the bit rate)
    int batch_chunk = 4096;  // images per fused-batch chunk (test hook: small values exercise the chunk loop)
    unsigned long long spin_ticks = 0;  // test hook: bound of the device-side waits between workgroups (100 MHz ticks, 0 = defaults)
    uint32_t fault = 0;      // test hook: fault injection mask (entropy.hip.h: EntropyLaunch::fault)
    bool profiling = false;
    int num_cus = 256;
    int k4_wgs_per_cu = 1;   // K4's grid = num_cus * this (create: what the device keeps resident)

    // device scratch (grown on demand, never shrunk)
    void* d_coef = nullptr;
    size_t coef_cap = 0;
    void* d_scan = nullptr;
    size_t scan_cap = 0;
    void* d_rgb = nullptr;
    size_t rgb_cap = 0;
    unsigned long long rgb_gen = 0;   // bumped by every call that writes pixels into d_rgb (kpeg_hip_resident_generation)
    uint32_t force_k0 = 0;      // debug key 8 / KPEG_FORCE_K0
    uint32_t fused_slots = 0;   // workgroups of k_sync_write resident at once; 0 = that kernel is not used (debug key 9 / KPEG_FUSED=0)
    uint32_t fused_slots_dev = 0;   // ... as the device reports it
    void* d_pad = nullptr;      // any-size extension: the padded picture K4 writes before the crop
    size_t pad_cap = 0;
    void* d_ebound = nullptr;  // per-block error bounds for K4 (written by K2 or k_ebound)
    size_t ebound_cap = 0;
    // compact coefficient stream between K2 and K4 (sparse streams whose MCU rows are whole K4 tiles): records, DC values, first record of every tile
    void* d_rec = nullptr;
    size_t rec_cap = 0;
    void* d_dc16 = nullptr;
    size_t dc16_cap = 0;
    void* d_tstart = nullptr;
    size_t tstart_cap = 0;
    int coef_layout = 0;       // test hook: 0 = chosen per call, 1 = always the dense layout, 2 = the compact stream wherever it is possible
    EntropyScratch ent;        // K0..K3 work buffers
    // [0] unused, [1] entropy error flag, [2] sync passes, [16..271] K4 exact-pixel counters
    uint32_t* d_status = nullptr;
    uint32_t* h_status = nullptr;  // pinned mirror: the last kernel of every call copies the device words to it (the error word [1] is sticky on the device until a sync has seen it), kpeg_hip_sync reads and clears it
    uint32_t* h_status_dev = nullptr;  // its device address
    bool status_clean = false;     // the device words are zero (the previous call's last kernel cleared them)

    enum { EV_BEGIN, EV_UNSTUFF, EV_SYNC, EV_SCAN, EV_WRITE, EV_DC, EV_IDCT, EV_COUNT };
    hipEvent_t ev[EV_COUNT] = {};
    hipEvent_t switch_ev = nullptr;   // kpeg_hip_set_stream: the new stream waits for what is queued on the old one
    bool ev_rec[EV_COUNT] = {};
    kpeg_hip_timings timings = {};
    bool status_pending = false;
    uint32_t status_seen[KPEG_STATUS_WORDS] = {};   // what the last kpeg_hip_sync read (test hook)

    // child contexts (stream + scratch each) for the batch entry points: the host-buffer batch alternates chunks
    // between two of them; a device batch outside the fused path's contract (restart markers inside the images)
    // goes round-robin over all
#ifndef KPEG_LANES
#define KPEG_LANES 6
#endif
    static const int NLANES = KPEG_LANES;
    kpeg_hip_ctx* lanes[NLANES] = {};
    hipEvent_t lane_ev[NLANES + 1] = {};   // [NLANES] = fork point on the parent's stream
    bool lanes_pending = false;            // parent: lanes hold deferred status
    bool keep_status = false;              // lane, during a batch: the device status words accumulate over the lane's images
    void* h_band[2] = {nullptr, nullptr};  // kpeg_hip_download_bands: pinned bounce buffers
    hipEvent_t h_band_ev[2] = {nullptr, nullptr};
    size_t h_band_cap = 0;
    void* h_scan = nullptr;                // lane: pinned staging for host-buffer batches
    void* d_batch = nullptr;               // fused batch: descriptor blob (pointer tables, lengths)
    size_t batch_cap = 0;
    void* h_batch[2] = {nullptr, nullptr}; // ... its pinned staging, double-buffered: a call only waits for the upload of the call before last
    hipEvent_t h_batch_ev[2] = {nullptr, nullptr};
    bool h_batch_busy[2] = {false, false};
    int h_batch_next = 0;
    size_t h_batch_cap = 0;
    size_t h_scan_cap = 0;
};

#define HIPCHK(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                  \
            return KPEG_HIP_E_DEVICE;                                                              \
        }                                                                                          \
    } while (0)

static int grow(kpeg_hip_ctx* ctx, void** p, size_t* cap, size_t need)
{
    if (need <= *cap) return KPEG_HIP_OK;
    if (*p) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    size_t want = need + need / 8 + 4096;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        ctx->last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
        return KPEG_HIP_E_NOMEM;
    }
    *cap = want;
    return KPEG_HIP_OK;
}

extern "C" int kpeg_hip_abi_version(void) { return KPEG_HIP_ABI_VERSION; }

#ifndef KPEG_SRC_HASH
#define KPEG_SRC_HASH "unstamped"
#endif
// libkpeg_amd/build.py compares this with the hash of the sources in the tree and rebuilds on a mismatch
extern "C" const char* kpeg_hip_build_hash(void)
{
    static const char stamp[] = "KPEG_SRC_HASH=" KPEG_SRC_HASH;
    return stamp + 14;
}

extern "C" const char* kpeg_hip_strerror(int code)
{
    switch (code) {
        case KPEG_HIP_OK: return "ok";
        case KPEG_HIP_E_ARG: return "invalid argument";
        case KPEG_HIP_E_DEVICE: return "HIP device/runtime error";
        case KPEG_HIP_E_TABLES: return "invalid Huffman table";
        case KPEG_HIP_E_STREAM: return "corrupt or truncated entropy-coded data";
        case KPEG_HIP_E_NOMEM: return "out of device memory";
        case KPEG_HIP_E_UNSUPPORTED: return "unsupported";
    }
    return "unknown error";
}

extern "C" const char* kpeg_hip_last_error(const kpeg_hip_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" int kpeg_hip_create(kpeg_hip_ctx** out, int device)
{
    if (!out) return KPEG_HIP_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return KPEG_HIP_E_DEVICE;
    kpeg_hip_ctx* ctx = new (std::nothrow) kpeg_hip_ctx;
    if (!ctx) return KPEG_HIP_E_NOMEM;
    ctx->device = device;
    auto fail = [&](const char* what, hipError_t e) {
        std::fprintf(stderr, "kpeg_hip_create: %s: %s\n", what, hipGetErrorString(e));
        delete ctx;
        return KPEG_HIP_E_DEVICE;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail("hipGetDeviceProperties", e);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "kpeg_hip_create: device %d is %s, this library is built for gfx950 only\n", device,
                     prop.gcnArchName);
        delete ctx;
        return KPEG_HIP_E_DEVICE;
    }
    ctx->num_cus = prop.multiProcessorCount;
    {
        // K4's workgroups are persistent workers, one per CU (K4_WAVES wavefronts each): launch exactly as many as stay
        // resident (more would run as a second, partly filled round).  KPEG_K4_WGS_PER_CU: timing experiments only.
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_idct_colour_fast<false>, K4_THREADS, 0) == hipSuccess && nb > 0) ctx->k4_wgs_per_cu = nb;
        if (const char* s = std::getenv("KPEG_K4_WGS_PER_CU")) {
            const int v = std::atoi(s);
            if (v > 0) ctx->k4_wgs_per_cu = v;
        }
        if (const char* s = std::getenv("KPEG_FORCE_K0")) ctx->force_k0 = std::atoi(s) ? 1u : 0u;
        {
            int nf = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, kpeg_dev::k_sync_write<kpeg_dev::SUBSEQ_SPARSE, false>, kpeg_dev::SYNC_WG, 0) == hipSuccess && nf > 0)
                ctx->fused_slots_dev = (uint32_t)nf * (uint32_t)ctx->num_cus;
            ctx->fused_slots = ctx->fused_slots_dev;
            if (const char* s = std::getenv("KPEG_FUSED")) ctx->fused_slots = std::atoi(s) ? ctx->fused_slots_dev : 0u;   // experiments: as debug key 9
        }
        if (const char* s = std::getenv("KPEG_COEF_LAYOUT")) ctx->coef_layout = std::atoi(s);   // experiments: as kpeg_hip_debug_set key 7
        if (std::getenv("KPEG_DEBUG")) std::fprintf(stderr, "kpeg_hip: K4 workgroups per CU: %d (occupancy query %d), %d wavefronts each\n", ctx->k4_wgs_per_cu, nb, K4_WAVES);
    }
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess)
        return fail("hipStreamCreate", e);
    ctx->stream = ctx->own_stream;
    for (int i = 0; i < kpeg_hip_ctx::EV_COUNT; ++i)
        if ((e = hipEventCreate(&ctx->ev[i])) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->switch_ev, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipMalloc((void**)&ctx->d_status, STATUS_BYTES)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipHostMalloc((void**)&ctx->h_status, STATUS_BYTES, hipHostMallocDefault)) != hipSuccess) return fail("hipHostMalloc", e);
    std::memset(ctx->h_status, 0, STATUS_BYTES);
    if ((e = hipHostGetDevicePointer((void**)&ctx->h_status_dev, ctx->h_status, 0)) != hipSuccess) return fail("hipHostGetDevicePointer", e);
    if ((e = hipMemset(ctx->d_status, 0, STATUS_BYTES)) != hipSuccess) return fail("hipMemset", e);
    ctx->status_clean = true;
    *out = ctx;
    return KPEG_HIP_OK;
}

extern "C" void kpeg_hip_destroy(kpeg_hip_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int l = 0; l < kpeg_hip_ctx::NLANES; ++l)
        if (ctx->lanes[l]) kpeg_hip_destroy(ctx->lanes[l]);
    for (int l = 0; l <= kpeg_hip_ctx::NLANES; ++l)
        if (ctx->lane_ev[l]) (void)hipEventDestroy(ctx->lane_ev[l]);
    for (int i = 0; i < 2; ++i) {
        if (ctx->h_band[i]) (void)hipHostFree(ctx->h_band[i]);
        if (ctx->h_band_ev[i]) (void)hipEventDestroy(ctx->h_band_ev[i]);
    }
    if (ctx->h_scan) (void)hipHostFree(ctx->h_scan);
    for (int i = 0; i < 2; ++i) {
        if (ctx->h_batch[i]) (void)hipHostFree(ctx->h_batch[i]);
        if (ctx->h_batch_ev[i]) (void)hipEventDestroy(ctx->h_batch_ev[i]);
    }
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    if (ctx->d_coef) (void)hipFree(ctx->d_coef);
    if (ctx->d_scan) (void)hipFree(ctx->d_scan);
    if (ctx->d_rgb) (void)hipFree(ctx->d_rgb);
    if (ctx->d_pad) (void)hipFree(ctx->d_pad);
    if (ctx->d_ebound) (void)hipFree(ctx->d_ebound);
    if (ctx->d_rec) (void)hipFree(ctx->d_rec);
    if (ctx->d_dc16) (void)hipFree(ctx->d_dc16);
    if (ctx->d_tstart) (void)hipFree(ctx->d_tstart);
    entropy_scratch_free(&ctx->ent);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    for (int i = 0; i < kpeg_hip_ctx::EV_COUNT; ++i)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->switch_ev) (void)hipEventDestroy(ctx->switch_ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int kpeg_hip_set_stream(kpeg_hip_ctx* ctx, void* s)
{
    if (!ctx) return KPEG_HIP_E_ARG;
    hipStream_t next = s ? (hipStream_t)s : ctx->own_stream;
    if (next == ctx->stream) return KPEG_HIP_OK;
    // The scratch buffers, the status words and the tables are ordered by the one stream only: work still queued
    // on the old stream must have finished before anything launched on the new one touches them.
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventRecord(ctx->switch_ev, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(next, ctx->switch_ev, 0));
    ctx->stream = next;
    return KPEG_HIP_OK;
}

extern "C" int kpeg_hip_set_profiling(kpeg_hip_ctx* ctx, int enable)
{
    if (!ctx) return KPEG_HIP_E_ARG;
    ctx->profiling = enable != 0;
    return KPEG_HIP_OK;
}

extern "C" int kpeg_hip_set_idct_mode(kpeg_hip_ctx* ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return KPEG_HIP_E_ARG;
    ctx->idct_mode = mode;
    return KPEG_HIP_OK;
}

static void mark(kpeg_hip_ctx* ctx, int which)
{
    if (!ctx->profiling) return;
    if (hipEventRecord(ctx->ev[which], ctx->stream) == hipSuccess) ctx->ev_rec[which] = true;
}

static void begin_call(kpeg_hip_ctx* ctx)
{
    for (int i = 0; i < kpeg_hip_ctx::EV_COUNT; ++i) ctx->ev_rec[i] = false;
    std::memset(&ctx->timings, 0, sizeof(ctx->timings));
}

extern "C" int kpeg_hip_sync(kpeg_hip_ctx* ctx)
{
    if (!ctx) return KPEG_HIP_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    int rc = KPEG_HIP_OK;
    if (ctx->lanes_pending) {
        // a batch: every lane's deferred status (errors are OR-ed over the lane's images)
        ctx->lanes_pending = false;
        uint32_t ex = 0;
        for (int l = 0; l < kpeg_hip_ctx::NLANES; ++l) {
            kpeg_hip_ctx* c = ctx->lanes[l];
            if (!c) continue;
            const int lrc = kpeg_hip_sync(c);
            c->keep_status = false;
            c->status_clean = false;   // the sums are still standing on the device
            ex += c->timings.exact_pixels;
            if (lrc && !rc) {
                rc = lrc;
                ctx->last_error = "batch: " + c->last_error;
            }
        }
        ctx->timings.exact_pixels = ex;
    }
    if (ctx->status_pending) {
        ctx->status_pending = false;
        uint32_t ex = 0;
        for (int i = 0; i < 256; ++i) ex += ctx->h_status[16 + i];
        ctx->timings.exact_pixels = ex;
        ctx->timings.sync_rounds = ctx->h_status[2];
        if (ctx->h_status[1] != 0) {
            // error flags of every call enqueued since the last sync (the device word is sticky: status_epilogue)
            if (ctx->h_status[1] & KPEG_ERR_TIMEOUT) {
                ctx->last_error = "a device-side wait between workgroups timed out (code " + std::to_string(ctx->h_status[1]) + ")";
                rc = KPEG_HIP_E_DEVICE;
            } else {
                ctx->last_error = "entropy decode flagged the stream as invalid (code " + std::to_string(ctx->h_status[1]) + ")";
                if (ctx->h_status[1] & 1u)
                    ctx->last_error += ": " + std::to_string(ctx->h_status[4]) + " restart segments found, the frame has " + std::to_string(ctx->h_status[5]);
                rc = KPEG_HIP_E_STREAM;
            }
            ctx->status_clean = false;   // the next call clears the device words before it starts
        }
        std::memcpy(ctx->status_seen, ctx->h_status, STATUS_BYTES);
        std::memset(ctx->h_status, 0, STATUS_BYTES);   // the stream is idle: nothing is adding to it
    }
    if (ctx->profiling) {
        auto span = [&](int a, int b) -> float {
            float ms = 0.f;
            if (ctx->ev_rec[a] && ctx->ev_rec[b] && hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]) == hipSuccess) return ms;
            return 0.f;
        };
        typedef kpeg_hip_ctx C;
        ctx->timings.unstuff_ms = span(C::EV_BEGIN, C::EV_UNSTUFF);
        ctx->timings.huff_sync_ms = span(C::EV_UNSTUFF, C::EV_SYNC);
        ctx->timings.huff_scan_ms = span(C::EV_SYNC, C::EV_SCAN);
        ctx->timings.huff_write_ms = span(C::EV_SCAN, C::EV_WRITE);
        ctx->timings.dc_ms = span(C::EV_WRITE, C::EV_DC);
        ctx->timings.idct_ms = ctx->ev_rec[C::EV_DC] ? span(C::EV_DC, C::EV_IDCT) : span(C::EV_BEGIN, C::EV_IDCT);
        ctx->timings.total_ms = span(C::EV_BEGIN, C::EV_IDCT);
        if (!ctx->ev_rec[C::EV_IDCT]) ctx->timings.total_ms = span(C::EV_BEGIN, C::EV_DC);
    }
    return rc;
}

extern "C" int kpeg_hip_get_timings(kpeg_hip_ctx* ctx, kpeg_hip_timings* out)
{
    if (!ctx || !out) return KPEG_HIP_E_ARG;
    *out = ctx->timings;
    return KPEG_HIP_OK;
}

// ---------------------------------------------------------------------------------------------
// any_size: the entry points of the any-size extension (whole-image decodes) take widths and heights that are not
// multiples of 8; everything else keeps the reference's contract.
static int check_frame(kpeg_hip_ctx* ctx, const kpeg_frame* f, bool any_size = false)
{
    if (!ctx || !f) return KPEG_HIP_E_ARG;
    if (f->width == 0 || f->height == 0 || (!any_size && ((f->width & 7) || (f->height & 7))) || f->width > 65535 || f->height > 65535) {
        ctx->last_error = any_size ? "width/height must be 1..65535" : "width/height must be non-zero multiples of 8 (SURVEY.md A.1)";
        return KPEG_HIP_E_ARG;
    }
    if (f->components != 0 && f->components != 1 && f->components != 3 && !(any_size && f->components == KPEG_FRAME_420)) {
        ctx->last_error = "components must be 0 / 3 (Y Cb Cr 4:4:4), 1 (grayscale) or, at the whole-image entry points, KPEG_FRAME_420";
        return KPEG_HIP_E_UNSUPPORTED;
    }
    return KPEG_HIP_OK;
}

static void natural_qtables(const kpeg_frame* f, QTables* qt)
{
    for (int t = 0; t < 2; ++t)
        for (int k = 0; k < 64; ++k) qt->q[t][KPEG_ZZ_TO_NATURAL[k]] = f->qt[t][k];
}

// K4 launch: rows [0, mcu_rows) of d_coef -> d_rgb.  ctx->d_ebound must hold the blocks' error bounds.
static int launch_idct(kpeg_hip_ctx* ctx, const kpeg_frame* f, const int16_t* d_coef, uint8_t* d_rgb, uint32_t mcu_rows,
                       uint8_t* const* d_rgb_table = nullptr, uint32_t rows_per_img = 0, bool compact = false)
{
    if ((!compact && (reinterpret_cast<uintptr_t>(d_coef) & 15)) || (reinterpret_cast<uintptr_t>(d_rgb) & 7)) {
        ctx->last_error = "device coefficient buffer must be 16-byte aligned, rgb buffer 8-byte aligned";
        return KPEG_HIP_E_ARG;
    }
    QTables qt;
    natural_qtables(f, &qt);
    IdctParams p;
    p.coef = d_coef;
    p.ebound = (const float*)ctx->d_ebound;
    p.rgb = d_rgb;
    p.mcus_w = f->width / 8;
    p.mcu_rows = mcu_rows;
    p.pitch = f->width * 3;
    p.stats = ctx->d_status + 16;
    p.status = ctx->d_status;
    p.h_status = ctx->idct_mode == 1 ? nullptr : ctx->h_status_dev;   // the per-MCU exact kernel has no epilogue
    p.keep_status = ctx->keep_status ? 1u : 0u;
    p.rgb_table = d_rgb_table;
    p.rows_per_img = rows_per_img;
    p.skip_exact = ctx->idct_mode == 2;
    p.rec = compact ? (const uint32_t*)ctx->d_rec : nullptr;
    p.dc16 = compact ? (const int16_t*)ctx->d_dc16 : nullptr;
    p.tile_start = compact ? (const uint32_t*)ctx->d_tstart : nullptr;
    p.rec_cap = compact ? (uint32_t)std::min<size_t>(ctx->rec_cap / 4, 0xFFFFFFFFu) : 0u;
    if (ctx->idct_mode == 1) {
        p.tiles_w = 0;
        p.ntiles = 0;
        p.tiles_w_magic = 0;
        p.tiles_w_shift = 0;
        hipLaunchKernelGGL(k_idct_colour_exact, dim3(p.mcus_w * mcu_rows), dim3(64), 0, ctx->stream, p, qt);
    } else {
        p.tiles_w = (p.mcus_w + TILE_MCUS - 1) / TILE_MCUS;
        p.ntiles = p.tiles_w * mcu_rows;
        {
            // exact x / d for 0 <= x < 2^31, d >= 2: m = ceil(2^(32+s) / d) with s = ceil(log2 d) - 1 fits 32
            // bits (2^s < d) and x * (m d - 2^(32+s)) < x d <= 2^(32+s) holds because 2^(s+1) >= d.
            // d == 1 is handled in the kernel.
            uint32_t d = p.tiles_w, lg = 0;
            while ((1u << lg) < d) ++lg;
            const uint32_t sft = lg ? lg - 1 : 0;
            p.tiles_w_magic = d > 1 ? (uint32_t)((((uint64_t)1 << (32 + sft)) + d - 1) / d) : 0;
            p.tiles_w_shift = sft;
        }
        const uint32_t resident = (uint32_t)ctx->num_cus * (uint32_t)ctx->k4_wgs_per_cu;
        const uint32_t want = (p.ntiles + K4_WAVES - 1) / K4_WAVES;   // at least a tile per wavefront
        const uint32_t grid = want < resident ? want : resident;
        if (compact) hipLaunchKernelGGL(k_idct_colour_fast<true>, dim3(grid), dim3(K4_THREADS), 0, ctx->stream, p, qt);
        else hipLaunchKernelGGL(k_idct_colour_fast<false>, dim3(grid), dim3(K4_THREADS), 0, ctx->stream, p, qt);
    }
    HIPCHK(ctx, hipGetLastError());
    return KPEG_HIP_OK;
}

__global__ void k_status_flush(uint32_t* status, uint32_t* h_status, uint32_t keep) { status_epilogue(status, h_status, 1, keep, 0); }

// End of an enqueued call: the status words reach the host mirror and the device words are zero again --
// done by K4's last wavefront when K4 was the call's last kernel, else by a one-wavefront kernel.
static int finish_async(kpeg_hip_ctx* ctx, bool k4_did_it)
{
    if (!k4_did_it) {
        hipLaunchKernelGGL(k_status_flush, dim3(1), dim3(64), 0, ctx->stream, ctx->d_status, ctx->h_status_dev, ctx->keep_status ? 1u : 0u);
        HIPCHK(ctx, hipGetLastError());
    }
    ctx->status_clean = !ctx->keep_status;
    ctx->status_pending = true;
    return KPEG_HIP_OK;
}

extern "C" int kpeg_hip_idct_colour_dev(kpeg_hip_ctx* ctx, const kpeg_frame* f, const int16_t* d_coef, uint8_t* d_rgb)
{
    int rc = check_frame(ctx, f);
    if (rc) return rc;
    if (!d_coef || !d_rgb) return KPEG_HIP_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    begin_call(ctx);
    const uint32_t nblocks = (f->width / 8) * (f->height / 8) * 3;
    if ((rc = grow(ctx, &ctx->d_ebound, &ctx->ebound_cap, (size_t)nblocks * sizeof(float)))) return rc;
    if (!ctx->status_clean && !ctx->keep_status) HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, STATUS_BYTES, ctx->stream));
    ctx->status_clean = false;
    mark(ctx, kpeg_hip_ctx::EV_BEGIN);
    {
        // caller-supplied coefficients carry no error bounds: derive them (K2 does this on the decode path)
        QTables qt;
        natural_qtables(f, &qt);
        hipLaunchKernelGGL(k_ebound, dim3((nblocks + EB_BLOCKS - 1) / EB_BLOCKS), dim3(256), 0, ctx->stream, d_coef, nblocks, qt,
                           (float*)ctx->d_ebound);
    }
    mark(ctx, kpeg_hip_ctx::EV_DC);
    rc = launch_idct(ctx, f, d_coef, d_rgb, f->height / 8);
    if (rc) return rc;
    mark(ctx, kpeg_hip_ctx::EV_IDCT);
    return finish_async(ctx, ctx->idct_mode != 1);
}

extern "C" int kpeg_hip_idct_colour(kpeg_hip_ctx* ctx, const kpeg_frame* f, const int16_t* coef, uint8_t* rgb)
{
    int rc = check_frame(ctx, f);
    if (rc) return rc;
    if (!coef || !rgb) return KPEG_HIP_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nmcu = (size_t)(f->width / 8) * (f->height / 8);
    const size_t cbytes = nmcu * 192 * sizeof(int16_t), rbytes = (size_t)f->width * f->height * 3;
    if ((rc = grow(ctx, &ctx->d_coef, &ctx->coef_cap, cbytes))) return rc;
    if ((rc = grow(ctx, &ctx->d_rgb, &ctx->rgb_cap, rbytes))) return rc;
    ctx->rgb_gen++;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_coef, coef, cbytes, hipMemcpyHostToDevice, ctx->stream));
    rc = kpeg_hip_idct_colour_dev(ctx, f, (const int16_t*)ctx->d_coef, (uint8_t*)ctx->d_rgb);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(rgb, ctx->d_rgb, rbytes, hipMemcpyDeviceToHost, ctx->stream));
    return kpeg_hip_sync(ctx);
}

// ---------------------------------------------------------------------------------------------
// Entropy decode + IDCT

// The compact coefficient stream between K2 and K4 (entropy.hip.h: WriteArgs) instead of the dense int16 layout: K1 clears
// nothing, K2 writes 4 bytes per non-zero AC coefficient in stream order instead of scattering 2-byte stores over 384 bytes
// per MCU, K4 reads a tenth of the bytes.  For sparse streams (the 96-bit sub-sequence path) whose MCU rows are whole K4
// tiles (width a multiple of 64), and never for the reference-order cross-check kernel, which reads dense blocks.
static bool want_compact(const kpeg_hip_ctx* ctx, const kpeg_frame* f, uint64_t scan_bytes, uint64_t nmcu)
{
    if (ctx->coef_layout == 1 || ctx->idct_mode == 1 || f->components == KPEG_FRAME_420) return false;   // (grayscale since the end of round 3: its one block per MCU takes the luma block's place in the stream, through the separate launches)
    if ((f->width / 8) % TILE_MCUS != 0) return false;
    const bool dense = entropy_dense_subseq(ctx->subseq, f->components == KPEG_FRAME_420, scan_bytes, nmcu);   // (the launcher's own rule)
    if (ctx->coef_layout == 2) return true;
    // Measured end to end (tools/layout_sizes.sh, round 2, us dense / compact): 1920x1080 (12 MiB of dense coefficients)
    // 91 / 93, 3840x2160 (48 MiB) 111 / 108, 7680x4320 (190 MiB) 220 / 199, 16384x16384 2050 / 1270, 256 x 1080p 141 -> 207
    // Gpixel/s: the dense layout's 2-byte scatter and its clear cost K2 and K1 more than the rebuild in LDS costs K4 once the
    // image is large enough for those to show beside the kernels' fixed latencies.
    // Dense streams (the 384-bit sub-sequences) too, up to 5 bits per pixel: an 8K photograph at 3.5 bit/px has ~25 non-zero terms
    // per block -- 100 bytes of records against 128 + 128 (clear) of dense coefficients -- and K2's consecutive 4-byte records cost
    // it 0.174 ms where the scattered 2-byte stores cost 0.223 (K4 0.127 against 0.110: the rebuild in LDS grows with the records);
    // the whole decode 0.591 -> 0.556 ms (round 3, profiles/r03_e).  Beyond that the records outgrow the blocks.
    // Small pictures (up to 32 MiB of dense coefficients), since the end of round 3: the compact stream where the sub-sequences are
    // the short ones -- it is what lets K1's pass 0 and K2 run as one kernel (k_sync_write), two launches less on pictures that are
    // all launch gaps and latency chains: 512x512 ... 2560x1440 synthetic 4-9 % faster, lena.jpg the same, a 640x424 photograph at
    // 3.5 bit/px 0.162 -> 0.145 ms; with the long sub-sequences (from 4 bits per pixel there) the dense layout stays ahead by 2 %
    // (tools/small_images_layout.py, profiles/r03_l_small_pictures_layout.txt).
    // (grayscale has no one kernel to gain there: k_write and K4 are 10-25 % slower on a 1080p picture with the records than with the
    // dense blocks, tools/experiments/gray_layout_check.py)
    if (nmcu * 384 <= ((uint64_t)32 << 20)) return !dense && f->components != 1;
    return !dense || scan_bytes * 8 < nmcu * 64 * 5;
}

static int run_entropy(kpeg_hip_ctx* ctx, const kpeg_frame* f, const uint8_t* d_scan, size_t scan_len, uint32_t nmcu,
                       int16_t* d_coef, const EntropyLaunch* batch = nullptr, bool compact = false)
{
    EntropyTables tabs;
    int rc = build_entropy_tables(f, &tabs);
    if (rc) {
        ctx->last_error = "Huffman table is not a usable prefix code";
        return KPEG_HIP_E_TABLES;
    }
    const bool sub420 = f->components == KPEG_FRAME_420;   // nmcu then counts 16x16 MCUs of six blocks
    if ((rc = grow(ctx, &ctx->d_ebound, &ctx->ebound_cap, (size_t)nmcu * (sub420 ? 6 : 3) * sizeof(float)))) return rc;
    EntropyLaunch L;
    L.stream = ctx->stream;
    L.d_scan = d_scan;
    L.scan_len = scan_len;
    L.nmcu = nmcu;
    L.restart_interval = f->restart_interval;
    L.d_coef = d_coef;
    L.d_ebound = (float*)ctx->d_ebound;
    L.d_status = ctx->d_status;
    L.num_cus = ctx->num_cus;
    L.sync_passes = ctx->sync_passes;
    L.warm = ctx->warm;
    L.subseq = ctx->subseq;
    L.gray = f->components == 1 ? 1u : 0u;
    L.sub420 = sub420 ? 1u : 0u;
    L.force_k0 = ctx->force_k0;
    L.fused_slots = ctx->fused_slots;
    if (compact) {
        // a record takes at least two bits of the stream (a one-bit code and a one-bit magnitude), a block holds at most 63
        const uint64_t bytes = batch ? batch->total_len : (uint64_t)scan_len;
        const uint64_t nrec = std::min<uint64_t>(bytes * 4, (uint64_t)nmcu * 3 * 63) + 64;
        if ((rc = grow(ctx, &ctx->d_rec, &ctx->rec_cap, nrec * 4))) return rc;
        if ((rc = grow(ctx, &ctx->d_dc16, &ctx->dc16_cap, (size_t)nmcu * 3 * 2 + 64))) return rc;
        if ((rc = grow(ctx, &ctx->d_tstart, &ctx->tstart_cap, ((size_t)nmcu / TILE_MCUS + 2) * 4))) return rc;
        L.d_rec = (uint32_t*)ctx->d_rec;
        L.rec_cap = (uint32_t)std::min<uint64_t>(nrec, 0xFFFFFFFFu);
        L.d_dc16 = (int16_t*)ctx->d_dc16;
        // grayscale: K2 writes the luma blocks' DC values only; the chroma blocks' (no records, bound "exact") read as zero
        if (f->components == 1) HIPCHK(ctx, hipMemsetAsync(ctx->d_dc16, 0, (size_t)nmcu * 3 * 2, ctx->stream));
        L.d_tile_start = (uint32_t*)ctx->d_tstart;
        L.ntiles = nmcu / TILE_MCUS;
    }
    L.spin_ticks = ctx->spin_ticks;
    L.fault = ctx->fault;
    if (batch) {
        L.nimg = batch->nimg;
        L.d_scan_tab = batch->d_scan_tab;
        L.d_len_tab = batch->d_len_tab;
        L.d_wg_tab = batch->d_wg_tab;
        L.total_parts = batch->total_parts;
        L.total_len = batch->total_len;
        L.restart_interval = batch->restart_interval;
    }
    hipEvent_t* evs = ctx->profiling ? ctx->ev : nullptr;
    rc = entropy_decode_launch(&ctx->ent, tabs, L, evs, ctx->ev_rec, &ctx->last_error);
    return rc;
}

extern "C" int kpeg_hip_entropy_decode_dev(kpeg_hip_ctx* ctx, const kpeg_frame* f, const uint8_t* d_scan, size_t scan_len,
                                           int16_t* d_coef)
{
    int rc = check_frame(ctx, f);
    if (rc) return rc;
    if (!d_scan || !scan_len || !d_coef) return KPEG_HIP_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    begin_call(ctx);
    if (!ctx->status_clean && !ctx->keep_status) HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, STATUS_BYTES, ctx->stream));
    ctx->status_clean = false;
    mark(ctx, kpeg_hip_ctx::EV_BEGIN);
    rc = run_entropy(ctx, f, d_scan, scan_len, (f->width / 8) * (f->height / 8), d_coef);
    if (rc) return rc;
    return finish_async(ctx, false);
}

// MCU rows [first_mcu_row, first_mcu_row + mcu_rows) of a picture of whole 8x8 MCUs
static int decode_stripe_whole_mcus(kpeg_hip_ctx* ctx, const kpeg_frame* f, const uint8_t* d_scan, size_t scan_len,
                                    uint32_t first_mcu_row, uint32_t mcu_rows, uint8_t* d_rgb)
{
    int rc = check_frame(ctx, f);
    if (rc) return rc;
    if (!d_scan || !scan_len || !d_rgb || mcu_rows == 0 || first_mcu_row + mcu_rows > f->height / 8) return KPEG_HIP_E_ARG;
    const uint32_t mw = f->width / 8;
    const bool whole = first_mcu_row == 0 && mcu_rows == f->height / 8;
    if (!whole) {
        // a stripe must start and end on restart-interval boundaries
        if (f->restart_interval == 0 || ((uint64_t)first_mcu_row * mw) % f->restart_interval ||
            (((uint64_t)mcu_rows * mw) % f->restart_interval && first_mcu_row + mcu_rows != f->height / 8)) {
            ctx->last_error = "stripe boundaries must coincide with restart intervals";
            return KPEG_HIP_E_ARG;
        }
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nmcu = (size_t)mw * mcu_rows;
    const bool compact = want_compact(ctx, f, scan_len, nmcu);
    if (!compact && (rc = grow(ctx, &ctx->d_coef, &ctx->coef_cap, nmcu * 192 * sizeof(int16_t)))) return rc;
    begin_call(ctx);
    if (!ctx->status_clean && !ctx->keep_status) HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, STATUS_BYTES, ctx->stream));
    ctx->status_clean = false;
    mark(ctx, kpeg_hip_ctx::EV_BEGIN);
    rc = run_entropy(ctx, f, d_scan, scan_len, (uint32_t)nmcu, (int16_t*)ctx->d_coef, nullptr, compact);
    if (rc) return rc;
    rc = launch_idct(ctx, f, (const int16_t*)ctx->d_coef, d_rgb, mcu_rows, nullptr, 0, compact);
    if (rc) return rc;
    mark(ctx, kpeg_hip_ctx::EV_IDCT);
    return finish_async(ctx, ctx->idct_mode != 1);
}

__global__ void k_crop(const uint8_t* __restrict__ src, uint32_t spitch, uint8_t* __restrict__ dst, uint32_t dpitch, uint64_t total);

// Any-size extension at the stripe entry: the stripe's MCU rows of the padded picture go to scratch, the rows and columns the picture
// has are packed into d_rgb (the stripe's first pixel row, rows of 3 * width bytes).
extern "C" int kpeg_hip_decode_stripe_dev(kpeg_hip_ctx* ctx, const kpeg_frame* f, const uint8_t* d_scan, size_t scan_len,
                                          uint32_t first_mcu_row, uint32_t mcu_rows, uint8_t* d_rgb)
{
    if (!ctx || !f) return KPEG_HIP_E_ARG;
    if (f->components == KPEG_FRAME_420) {
        ctx->last_error = "4:2:0 pictures are decoded whole (kpeg_hip_decode_scan*, kpeg_hip_decode_batch*), not in stripes";
        return KPEG_HIP_E_UNSUPPORTED;
    }
    if (!((f->width & 7) || (f->height & 7))) return decode_stripe_whole_mcus(ctx, f, d_scan, scan_len, first_mcu_row, mcu_rows, d_rgb);
    int rc = check_frame(ctx, f, true);
    if (rc) return rc;
    kpeg_frame fp = *f;
    fp.width = (f->width + 7) & ~7u;
    fp.height = (f->height + 7) & ~7u;
    if (!d_scan || !scan_len || !d_rgb || mcu_rows == 0 || first_mcu_row + mcu_rows > fp.height / 8) return KPEG_HIP_E_ARG;
    if (reinterpret_cast<uintptr_t>(d_rgb) & 7) {
        ctx->last_error = "rgb buffer must be 8-byte aligned";
        return KPEG_HIP_E_ARG;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = grow(ctx, &ctx->d_pad, &ctx->pad_cap, (size_t)fp.width * 3 * mcu_rows * 8))) return rc;
    rc = decode_stripe_whole_mcus(ctx, &fp, d_scan, scan_len, first_mcu_row, mcu_rows, (uint8_t*)ctx->d_pad);
    if (rc) return rc;
    const uint32_t out_rows = std::min(mcu_rows * 8, f->height - first_mcu_row * 8);
    const uint64_t total = (uint64_t)f->width * 3 * out_rows;
    hipLaunchKernelGGL(k_crop, dim3((unsigned)((total / 4 + 256) / 256)), dim3(256), 0, ctx->stream, (const uint8_t*)ctx->d_pad, fp.width * 3,
                       d_rgb, f->width * 3, total);
    HIPCHK(ctx, hipGetLastError());
    return KPEG_HIP_OK;
}

// Any-size extension (Image::createImageFromMCUs, Image.cpp:26-27,73-84: pad to whole MCUs, tile, pop the extra columns
// and rows): rows [0, H) x bytes [0, 3 W) of the padded picture, four destination bytes per thread.
__global__ void k_crop(const uint8_t* __restrict__ src, uint32_t spitch, uint8_t* __restrict__ dst, uint32_t dpitch, uint64_t total)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;
    uint32_t y = (uint32_t)(i / dpitch), x = (uint32_t)(i - (uint64_t)y * dpitch), w = 0;
    const uint32_t nb = (uint32_t)(total - i < 4 ? total - i : 4);
    if (nb == 4 && x + 4 <= dpitch) {
        // the four bytes lie in one row: two aligned words of the source and a funnel shift instead of four byte loads (the word
        // behind a row's last may be the next row's or the buffer's slack: read, and shifted out) -- 12 Mpixel: 34 -> 21 us
        const size_t s = (size_t)y * spitch + x;
        const uint32_t* const s32 = reinterpret_cast<const uint32_t*>(src + (s & ~(size_t)3));   // (src is 16-byte aligned: a context's own scratch)
        const uint32_t lo = s32[0], hi = (s & 3) ? s32[1] : 0u;
        *reinterpret_cast<uint32_t*>(dst + i) = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(s & 3));   // (dst is 8-byte aligned: launch_idct's rule for every destination)
        return;
    }
    for (uint32_t k = 0; k < nb; ++k) {
        if (x == dpitch) {
            x = 0;
            y++;
        }
        w |= (uint32_t)src[(size_t)y * spitch + x] << (8 * k);
        x++;
    }
    if (nb == 4) {
        *reinterpret_cast<uint32_t*>(dst + i) = w;
    } else {
        for (uint32_t k = 0; k < nb; ++k) dst[i + k] = (uint8_t)(w >> (8 * k));
    }
}

static int decode_any_size(kpeg_hip_ctx* ctx, const kpeg_frame* f, const uint8_t* d_scan, size_t scan_len, uint8_t* d_rgb)
{
    return kpeg_hip_decode_stripe_dev(ctx, f, d_scan, scan_len, 0, (f->height + 7) / 8, d_rgb);
}

// 4:2:0 extension: K0-K2 with six blocks per MCU, k_idct_colour_fast_420 on 16x16 MCUs into the padded picture (idct mode 1: the
// reference-order kernel, every sample evaluated as MCU::computeIDCT would), crop.
static int decode_420(kpeg_hip_ctx* ctx, const kpeg_frame* f, const uint8_t* d_scan, size_t scan_len, uint8_t* d_rgb)
{
    int rc = check_frame(ctx, f, true);
    if (rc) return rc;
    if (!d_scan || !scan_len || !d_rgb) return KPEG_HIP_E_ARG;
    if (reinterpret_cast<uintptr_t>(d_rgb) & 7) {
        ctx->last_error = "rgb buffer must be 8-byte aligned";
        return KPEG_HIP_E_ARG;
    }
    const uint32_t mw = (f->width + 15) / 16, mh = (f->height + 15) / 16;
    const size_t nmcu = (size_t)mw * mh;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = grow(ctx, &ctx->d_coef, &ctx->coef_cap, nmcu * 384 * sizeof(int16_t)))) return rc;
    if ((rc = grow(ctx, &ctx->d_pad, &ctx->pad_cap, nmcu * 768))) return rc;
    begin_call(ctx);
    if (!ctx->status_clean && !ctx->keep_status) HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, STATUS_BYTES, ctx->stream));
    ctx->status_clean = false;
    mark(ctx, kpeg_hip_ctx::EV_BEGIN);
    rc = run_entropy(ctx, f, d_scan, scan_len, (uint32_t)nmcu, (int16_t*)ctx->d_coef);
    if (rc) return rc;
    QTables qt;
    natural_qtables(f, &qt);
    // a picture of whole MCUs needs no crop: the pixel kernel writes the caller's buffer
    const bool whole = (f->width & 15) == 0 && (f->height & 15) == 0;
    uint8_t* const out = whole ? d_rgb : (uint8_t*)ctx->d_pad;
    if (ctx->idct_mode == 1) {
        // cross-check: every sample in the reference's order
        hipLaunchKernelGGL(k_idct_colour_exact_420, dim3((unsigned)nmcu), dim3(256), 0, ctx->stream, (const int16_t*)ctx->d_coef, out, mw, mw * 48, qt);
    } else {
        Idct420Params p;
        p.coef = (const int16_t*)ctx->d_coef, p.ebound = (const float*)ctx->d_ebound, p.rgb = out;
        p.mcus_w = mw, p.mcus_h = mh, p.pitch = mw * 48, p.tiles_w = (mw + 3) / 4, p.ntiles = p.tiles_w * mh;
        p.stats = ctx->d_status + 16;
        hipLaunchKernelGGL(k_idct_colour_fast_420, dim3((p.ntiles + 3) / 4), dim3(256), 0, ctx->stream, p, qt);
    }
    if (!whole) {
        const uint64_t total = (uint64_t)f->width * f->height * 3;
        hipLaunchKernelGGL(k_crop, dim3((unsigned)((total / 4 + 256) / 256)), dim3(256), 0, ctx->stream, (const uint8_t*)ctx->d_pad, mw * 48, d_rgb,
                           f->width * 3, total);
    }
    HIPCHK(ctx, hipGetLastError());
    mark(ctx, kpeg_hip_ctx::EV_IDCT);
    return finish_async(ctx, false);
}

extern "C" int kpeg_hip_decode_scan_dev(kpeg_hip_ctx* ctx, const kpeg_frame* f, const uint8_t* d_scan, size_t scan_len,
                                        uint8_t* d_rgb)
{
    if (!f) return KPEG_HIP_E_ARG;
    if (f->components == KPEG_FRAME_420) return decode_420(ctx, f, d_scan, scan_len, d_rgb);
    if ((f->width & 7) || (f->height & 7)) return decode_any_size(ctx, f, d_scan, scan_len, d_rgb);
    return kpeg_hip_decode_stripe_dev(ctx, f, d_scan, scan_len, 0, f->height / 8, d_rgb);
}

extern "C" int kpeg_hip_decode_scan(kpeg_hip_ctx* ctx, const kpeg_frame* f, const uint8_t* scan, size_t scan_len, uint8_t* rgb)
{
    int rc = check_frame(ctx, f, true);
    if (rc) return rc;
    if (!scan || !scan_len || !rgb) return KPEG_HIP_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t rbytes = (size_t)f->width * f->height * 3;
    if ((rc = grow(ctx, &ctx->d_scan, &ctx->scan_cap, scan_len + 64))) return rc;
    if ((rc = grow(ctx, &ctx->d_rgb, &ctx->rgb_cap, rbytes))) return rc;
    ctx->rgb_gen++;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_scan, scan, scan_len, hipMemcpyHostToDevice, ctx->stream));
    rc = kpeg_hip_decode_scan_dev(ctx, f, (const uint8_t*)ctx->d_scan, scan_len, (uint8_t*)ctx->d_rgb);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(rgb, ctx->d_rgb, rbytes, hipMemcpyDeviceToHost, ctx->stream));
    return kpeg_hip_sync(ctx);
}

extern "C" int kpeg_hip_decode_scan_resident(kpeg_hip_ctx* ctx, const kpeg_frame* f, const uint8_t* scan, size_t scan_len)
{
    int rc = check_frame(ctx, f, true);
    if (rc) return rc;
    if (!scan || !scan_len) return KPEG_HIP_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t rbytes = (size_t)f->width * f->height * 3;
    if ((rc = grow(ctx, &ctx->d_scan, &ctx->scan_cap, scan_len + 64))) return rc;
    if ((rc = grow(ctx, &ctx->d_rgb, &ctx->rgb_cap, rbytes))) return rc;
    ctx->rgb_gen++;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_scan, scan, scan_len, hipMemcpyHostToDevice, ctx->stream));
    rc = kpeg_hip_decode_scan_dev(ctx, f, (const uint8_t*)ctx->d_scan, scan_len, (uint8_t*)ctx->d_rgb);
    if (rc) return rc;
    return kpeg_hip_sync(ctx);
}

extern "C" unsigned long long kpeg_hip_resident_generation(const kpeg_hip_ctx* ctx) { return ctx ? ctx->rgb_gen : 0ull; }

extern "C" int kpeg_hip_download_bands(kpeg_hip_ctx* ctx, const kpeg_frame* f, uint32_t band_rows, kpeg_hip_band_sink sink, void* user)
{
    int rc = check_frame(ctx, f, true);
    if (rc) return rc;
    if (!sink) return KPEG_HIP_E_ARG;
    const size_t pitch = (size_t)f->width * 3, rbytes = pitch * f->height;
    if (!ctx->d_rgb || ctx->rgb_cap < rbytes) {
        ctx->last_error = "no decoded image of this size is resident (kpeg_hip_decode_scan_resident first)";
        return KPEG_HIP_E_ARG;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (band_rows == 0) band_rows = (uint32_t)std::max<size_t>(8, (((size_t)8 << 20) / pitch) & ~(size_t)7);
    band_rows = std::min(band_rows, f->height);
    const size_t bbytes = pitch * band_rows;
    if (bbytes > ctx->h_band_cap) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 2; ++i) {
            if (ctx->h_band[i]) (void)hipHostFree(ctx->h_band[i]);
            ctx->h_band[i] = nullptr;
        }
        ctx->h_band_cap = 0;
        for (int i = 0; i < 2; ++i) HIPCHK(ctx, hipHostMalloc(&ctx->h_band[i], bbytes, hipHostMallocDefault));
        ctx->h_band_cap = bbytes;
    }
    for (int i = 0; i < 2; ++i)
        if (!ctx->h_band_ev[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->h_band_ev[i], hipEventDisableTiming));
    const uint32_t nb = (f->height + band_rows - 1) / band_rows;
    auto rows_of = [&](uint32_t k) { return std::min(band_rows, f->height - k * band_rows); };
    auto issue = [&](uint32_t k) -> hipError_t {
        hipError_t e = hipMemcpyAsync(ctx->h_band[k & 1], (const uint8_t*)ctx->d_rgb + (size_t)k * band_rows * pitch, pitch * rows_of(k),
                                      hipMemcpyDeviceToHost, ctx->stream);
        return e != hipSuccess ? e : hipEventRecord(ctx->h_band_ev[k & 1], ctx->stream);
    };
    auto deliver = [&](uint32_t k) -> int {
        if (hipEventSynchronize(ctx->h_band_ev[k & 1]) != hipSuccess) return KPEG_HIP_E_DEVICE;
        return sink(user, k * band_rows, rows_of(k), (const uint8_t*)ctx->h_band[k & 1], pitch * rows_of(k)) ? KPEG_HIP_E_ARG : KPEG_HIP_OK;
    };
    HIPCHK(ctx, issue(0));
    for (uint32_t k = 1; k < nb; ++k) {
        HIPCHK(ctx, issue(k));                   // band k crosses PCIe ...
        if ((rc = deliver(k - 1))) break;        // ... while the sink has band k - 1
    }
    if (!rc) rc = deliver(nb - 1);
    (void)hipStreamSynchronize(ctx->stream);
    if (rc == KPEG_HIP_E_ARG) ctx->last_error = "the band sink stopped the download";
    return rc;
}

static int ensure_lanes(kpeg_hip_ctx* ctx)
{
    for (int l = 0; l < kpeg_hip_ctx::NLANES; ++l) {
        if (!ctx->lanes[l]) {
            int rc = kpeg_hip_create(&ctx->lanes[l], ctx->device);
            if (rc) {
                ctx->last_error = "batch lane: kpeg_hip_create failed";
                return rc;
            }
        }
        ctx->lanes[l]->idct_mode = ctx->idct_mode;
        ctx->lanes[l]->sync_passes = ctx->sync_passes;
        ctx->lanes[l]->warm = ctx->warm;
        ctx->lanes[l]->subseq = ctx->subseq;
        ctx->lanes[l]->coef_layout = ctx->coef_layout;
        ctx->lanes[l]->fused_slots = ctx->fused_slots ? ctx->lanes[l]->fused_slots_dev : 0u;
    }
    for (int l = 0; l <= kpeg_hip_ctx::NLANES; ++l)
        if (!ctx->lane_ev[l]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->lane_ev[l], hipEventDisableTiming));
    return KPEG_HIP_OK;
}

// lanes start behind whatever is already queued on the parent's stream ...
static int lanes_fork(kpeg_hip_ctx* ctx)
{
    HIPCHK(ctx, hipEventRecord(ctx->lane_ev[kpeg_hip_ctx::NLANES], ctx->stream));
    for (int l = 0; l < kpeg_hip_ctx::NLANES; ++l) {
        HIPCHK(ctx, hipStreamWaitEvent(ctx->lanes[l]->stream, ctx->lane_ev[kpeg_hip_ctx::NLANES], 0));
        kpeg_hip_ctx* c = ctx->lanes[l];
        if (!c->status_clean) HIPCHK(ctx, hipMemsetAsync(c->d_status, 0, STATUS_BYTES, c->stream));
        c->status_clean = true;
        c->keep_status = true;   // error flags and counters of the lane's images add up on the device
    }
    return KPEG_HIP_OK;
}

// ... and the parent's stream continues behind all of them
static int lanes_join(kpeg_hip_ctx* ctx)
{
    for (int l = 0; l < kpeg_hip_ctx::NLANES; ++l) {
        HIPCHK(ctx, hipEventRecord(ctx->lane_ev[l], ctx->lanes[l]->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_ev[l], 0));
    }
    ctx->lanes_pending = true;
    return KPEG_HIP_OK;
}

// Fused batch: the images of a chunk are decoded as the restart segments of one virtual stream -- every
// segment starts from the known state with its DC predictors reset, which is exactly what an independent
// image needs -- so one set of launches covers them all (K0 gathers the scans through a pointer table,
// K4 scatters the rows through another).  Launch gaps and half-empty grids are what small images cost on
// their own; fused, a batch of 1080p images runs at the rate of one large image.
static int decode_batch_fused(kpeg_hip_ctx* ctx, int count, const kpeg_frame* f, const uint8_t* const* d_scans, const size_t* scan_lens,
                              uint8_t* const* d_rgbs)
{
    const uint32_t nmcu1 = (f->width / 8) * (f->height / 8);
    int done = 0;
    while (done < count) {
        // a chunk: as many images as fit 256 MiB of scan data, 2^27 blocks and 4096 segments
        int n = 0;
        uint64_t bytes = 0;
        while (done + n < count && n < ctx->batch_chunk && (uint64_t)(n + 1) * nmcu1 * 3 < (1ull << 27)) {
            const size_t len = scan_lens[done + n];
            if (!d_scans[done + n] || !d_rgbs[done + n] || len == 0) return KPEG_HIP_E_ARG;
            if ((reinterpret_cast<uintptr_t>(d_rgbs[done + n]) & 15) || len >= (1ull << 28)) {
                ctx->last_error = "batch: rgb buffers must be 16-byte aligned, scans below 256 MiB";
                return KPEG_HIP_E_ARG;
            }
            if (bytes + len >= (1ull << 28)) break;
            bytes += len;
            ++n;
        }
        if (n == 0) {
            ctx->last_error = "batch: image too large for the fused path";
            return KPEG_HIP_E_UNSUPPORTED;
        }
        // descriptor blob: [n] scan pointers, [n] rgb pointers, [n] lengths, [n + 1] first K0 workgroup
        const size_t blob = (size_t)n * 16 + (size_t)n * 4 + (size_t)(n + 1) * 4;
        int rc;
        if ((rc = grow(ctx, &ctx->d_batch, &ctx->batch_cap, blob))) return rc;
        if (blob > ctx->h_batch_cap) {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            for (int i = 0; i < 2; ++i) {
                if (ctx->h_batch[i]) (void)hipHostFree(ctx->h_batch[i]);
                ctx->h_batch[i] = nullptr;
                ctx->h_batch_busy[i] = false;
            }
            ctx->h_batch_cap = 0;
            for (int i = 0; i < 2; ++i) HIPCHK(ctx, hipHostMalloc(&ctx->h_batch[i], blob * 2, hipHostMallocDefault));
            ctx->h_batch_cap = blob * 2;
        }
        const int hbi = ctx->h_batch_next;
        ctx->h_batch_next ^= 1;
        if (!ctx->h_batch_ev[hbi]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->h_batch_ev[hbi], hipEventDisableTiming));
        // the upload that last used this staging buffer (two chunks ago) has left it: a wait for that copy only,
        // not for the decodes queued since
        if (ctx->h_batch_busy[hbi]) HIPCHK(ctx, hipEventSynchronize(ctx->h_batch_ev[hbi]));
        uint8_t* hb = (uint8_t*)ctx->h_batch[hbi];
        const uint8_t** h_scan = (const uint8_t**)hb;
        uint8_t** h_rgb = (uint8_t**)(hb + (size_t)n * 8);
        uint32_t* h_len = (uint32_t*)(hb + (size_t)n * 16);
        uint32_t* h_wg = h_len + n;
        uint32_t parts = 0;
        for (int i = 0; i < n; ++i) {
            h_scan[i] = d_scans[done + i];
            h_rgb[i] = d_rgbs[done + i];
            h_len[i] = (uint32_t)scan_lens[done + i];
            h_wg[i] = parts;
            parts += (h_len[i] + US_BLOCK_BYTES - 1) / US_BLOCK_BYTES;
        }
        h_wg[n] = parts;
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_batch, hb, blob, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->h_batch_ev[hbi], ctx->stream));
        ctx->h_batch_busy[hbi] = true;
        const uint8_t* db = (const uint8_t*)ctx->d_batch;
        EntropyLaunch B;
        B.nimg = (uint32_t)n;
        B.d_scan_tab = (const uint8_t* const*)db;
        B.d_len_tab = (const uint32_t*)(db + (size_t)n * 16);
        B.d_wg_tab = B.d_len_tab + n;
        B.total_parts = parts;
        B.total_len = bytes;
        B.restart_interval = nmcu1;
        const size_t nmcu = (size_t)nmcu1 * n;
        const bool compact = want_compact(ctx, f, bytes, nmcu);
        if (!compact && (rc = grow(ctx, &ctx->d_coef, &ctx->coef_cap, nmcu * 192 * sizeof(int16_t)))) return rc;
        begin_call(ctx);
        if (!ctx->status_clean) HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, STATUS_BYTES, ctx->stream));
        ctx->status_clean = false;
        rc = run_entropy(ctx, f, nullptr, 0, (uint32_t)nmcu, (int16_t*)ctx->d_coef, &B, compact);
        if (rc) return rc;
        rc = launch_idct(ctx, f, (const int16_t*)ctx->d_coef, nullptr, (uint32_t)(f->height / 8) * n, (uint8_t* const*)(db + (size_t)n * 8),
                         f->height / 8, compact);
        if (rc) return rc;
        if ((rc = finish_async(ctx, true))) return rc;
        done += n;
        if (done < count) {
            // the next chunk reuses the scratch and would overwrite the status words: settle this one first
            if ((rc = kpeg_hip_sync(ctx))) return rc;
        }
    }
    return KPEG_HIP_OK;
}

extern "C" int kpeg_hip_decode_batch_dev(kpeg_hip_ctx* ctx, int count, const kpeg_frame* f, const uint8_t* const* d_scans,
                                         const size_t* scan_lens, uint8_t* const* d_rgbs)
{
    int rc = check_frame(ctx, f, true);
    if (rc) return rc;
    if (count <= 0 || !d_scans || !scan_lens || !d_rgbs) return KPEG_HIP_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool odd = (f->width & 7) || (f->height & 7), s420 = f->components == KPEG_FRAME_420;
    if (!f->restart_interval && ctx->idct_mode != 1 && !s420) {
        if (!odd) return decode_batch_fused(ctx, count, f, d_scans, scan_lens, d_rgbs);
        // any-size extension: the padded pictures through the fused path into scratch, up to 1 GiB of them at a time, then the
        // rows and columns the pictures have are packed into the callers' buffers
        kpeg_frame fp = *f;
        fp.width = (f->width + 7) & ~7u;
        fp.height = (f->height + 7) & ~7u;
        const size_t pb = (size_t)fp.width * fp.height * 3;   // (a multiple of 16: the fused path's alignment rule holds for every picture)
        const int per = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, ((size_t)1 << 30) / pb));
        const uint64_t total = (uint64_t)f->width * f->height * 3;
        for (int done = 0; done < count; done += per) {
            const int n = std::min(per, count - done);
            for (int i = 0; i < n; ++i)
                if (!d_rgbs[done + i] || (reinterpret_cast<uintptr_t>(d_rgbs[done + i]) & 7)) {
                    ctx->last_error = "batch: rgb buffers must be 8-byte aligned";
                    return KPEG_HIP_E_ARG;
                }
            if (done && (rc = kpeg_hip_sync(ctx))) return rc;   // the scratch is the previous pictures' until their crops have run
            if ((rc = grow(ctx, &ctx->d_pad, &ctx->pad_cap, pb * n))) return rc;
            std::vector<uint8_t*> pads(n);
            for (int i = 0; i < n; ++i) pads[i] = (uint8_t*)ctx->d_pad + pb * i;
            if ((rc = decode_batch_fused(ctx, n, &fp, d_scans + done, scan_lens + done, pads.data()))) return rc;
            for (int i = 0; i < n; ++i)
                hipLaunchKernelGGL(k_crop, dim3((unsigned)((total / 4 + 256) / 256)), dim3(256), 0, ctx->stream, (const uint8_t*)pads[i], fp.width * 3,
                                   d_rgbs[done + i], f->width * 3, total);
            HIPCHK(ctx, hipGetLastError());
        }
        return KPEG_HIP_OK;
    }
    // restart markers inside the pictures, 4:2:0, the reference-order kernel: picture by picture, round-robin over the lanes
    if ((rc = ensure_lanes(ctx))) return rc;
    if ((rc = lanes_fork(ctx))) return rc;
    for (int i = 0; i < count; ++i) {
        kpeg_hip_ctx* c = ctx->lanes[i % kpeg_hip_ctx::NLANES];
        rc = kpeg_hip_decode_scan_dev(c, f, d_scans[i], scan_lens[i], d_rgbs[i]);
        if (rc) {
            ctx->last_error = "batch image " + std::to_string(i) + ": " + c->last_error;
            (void)lanes_join(ctx);
            return rc;
        }
    }
    return lanes_join(ctx);
}

extern "C" int kpeg_hip_decode_batch(kpeg_hip_ctx* ctx, int count, const kpeg_frame* f, const uint8_t* const* scans,
                                     const size_t* scan_lens, uint8_t* const* rgbs)
{
    int rc = check_frame(ctx, f, true);
    if (rc) return rc;
    if (count <= 0 || !scans || !scan_lens || !rgbs) return KPEG_HIP_E_ARG;
    for (int i = 0; i < count; ++i)
        if (!scans[i] || !scan_lens[i] || !rgbs[i]) return KPEG_HIP_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (f->restart_interval || ctx->idct_mode == 1 || (f->width & 7) || (f->height & 7) || f->components == KPEG_FRAME_420) {
        // outside the fused path's contract: one by one
        for (int i = 0; i < count; ++i)
            if ((rc = kpeg_hip_decode_scan(ctx, f, scans[i], scan_lens[i], rgbs[i]))) return rc;
        return KPEG_HIP_OK;
    }
    if ((rc = ensure_lanes(ctx))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // Chunks of images alternate between two lanes (stream + buffers + pinned staging each): a chunk's scans
    // go up in one copy and are decoded by the fused batch path; while the host sits in the (blocking)
    // downloads of the previous chunk into the caller's pageable buffers, the GPU decodes this one.
    const size_t rbytes = (size_t)f->width * f->height * 3;
    const int per_chunk = (int)std::max<size_t>(1, std::min<size_t>(64, ((size_t)192 << 20) / rbytes));
    struct Chunk { int first = 0, n = 0; };
    Chunk prev;
    bool have_prev = false;
    auto download = [&](const Chunk& c, int lane) -> int {
        kpeg_hip_ctx* L = ctx->lanes[lane];
        int r = kpeg_hip_sync(L);   // decode finished, status checked
        if (r) {
            ctx->last_error = "batch images " + std::to_string(c.first) + ".." + std::to_string(c.first + c.n - 1) + ": " + L->last_error;
            return r;
        }
        for (int i = 0; i < c.n; ++i)
            HIPCHK(ctx, hipMemcpyAsync(rgbs[c.first + i], (uint8_t*)L->d_rgb + (size_t)i * rbytes, rbytes, hipMemcpyDeviceToHost, L->stream));
        HIPCHK(ctx, hipStreamSynchronize(L->stream));
        return KPEG_HIP_OK;
    };
    int lane = 0;
    for (int first = 0; first < count; first += per_chunk, lane ^= 1) {
        Chunk c;
        c.first = first;
        c.n = std::min(per_chunk, count - first);
        kpeg_hip_ctx* L = ctx->lanes[lane];
        L->idct_mode = ctx->idct_mode;
        size_t total = 0;
        std::vector<size_t> off(c.n);
        for (int i = 0; i < c.n; ++i) {
            off[i] = total;
            total += (scan_lens[first + i] + 63) & ~(size_t)63;
        }
        if ((rc = grow(L, &L->d_scan, &L->scan_cap, total + 64))) return rc;
        if ((rc = grow(L, &L->d_rgb, &L->rgb_cap, rbytes * c.n))) return rc;
        L->rgb_gen++;
        if (total > L->h_scan_cap) {
            if (L->h_scan) (void)hipHostFree(L->h_scan);
            L->h_scan = nullptr;
            L->h_scan_cap = 0;
            HIPCHK(ctx, hipHostMalloc(&L->h_scan, total + total / 4 + 4096, hipHostMallocDefault));
            L->h_scan_cap = total + total / 4 + 4096;
        }
        std::vector<const uint8_t*> dsc(c.n);
        std::vector<uint8_t*> drg(c.n);
        for (int i = 0; i < c.n; ++i) {
            std::memcpy((uint8_t*)L->h_scan + off[i], scans[first + i], scan_lens[first + i]);
            dsc[i] = (const uint8_t*)L->d_scan + off[i];
            drg[i] = (uint8_t*)L->d_rgb + (size_t)i * rbytes;
        }
        HIPCHK(ctx, hipMemcpyAsync(L->d_scan, L->h_scan, total, hipMemcpyHostToDevice, L->stream));
        rc = decode_batch_fused(L, c.n, f, dsc.data(), scan_lens + first, drg.data());
        if (rc) {
            ctx->last_error = "batch images " + std::to_string(first) + "..: " + L->last_error;
            return rc;
        }
        if (have_prev && (rc = download(prev, lane ^ 1))) return rc;
        prev = c;
        have_prev = true;
    }
    return download(prev, lane ^ 1);
}

// ---------------------------------------------------------------------------------------------
// One image over several GPUs, single host thread (include/kpeg_hip.h: kpeg_hip_decode_sharded).
static int decode_sharded_impl(kpeg_hip_ctx* const* ctxs, int ngpu, const kpeg_frame* f, const uint8_t* scan, size_t scan_len,
                               uint8_t* rgb_root, bool root_is_device)
{
    if (!ctxs || ngpu <= 0 || !ctxs[0]) return KPEG_HIP_E_ARG;
    kpeg_hip_ctx* root = ctxs[0];
    int rc = check_frame(root, f);
    if (rc) return rc;
    if (!scan || !scan_len || !rgb_root) return KPEG_HIP_E_ARG;
    for (int g = 0; g < ngpu; ++g)
        if (!ctxs[g]) return KPEG_HIP_E_ARG;
    const uint32_t mw = f->width / 8, mh = f->height / 8;
    if (ngpu > 1 && f->restart_interval == 0) {
        root->last_error = "a stream without restart markers is one serial bit string: it cannot be sharded";
        return KPEG_HIP_E_UNSUPPORTED;
    }
    // stripes of whole MCU rows; every stripe must start on a restart-interval boundary
    const uint32_t rows_per = (mh + (uint32_t)ngpu - 1) / (uint32_t)ngpu;
    struct Stripe { uint32_t r0 = 0, nr = 0; size_t b0 = 0, b1 = 0; };
    std::vector<Stripe> st((size_t)ngpu);
    if (ngpu == 1) {
        st[0].r0 = 0, st[0].nr = mh, st[0].b0 = 0, st[0].b1 = scan_len;
    } else {
        // RSTn positions in the still-stuffed scan: FF D0..D7 cannot occur inside entropy-coded data.  memchr finds the FF bytes
        // (one in ~200 bytes of a q75 stream) at memory speed: 35 MB of the 16384 x 16384 image in a few ms, where a byte loop took tens.
        const uint64_t nint = ((uint64_t)mw * mh + f->restart_interval - 1) / f->restart_interval;
        // ... and in slices, one host thread each (the 35 MB are 3 ms for one thread -- longer than eight GPUs take to decode them): a marker
        // belongs to the slice its FF lies in (the byte behind it may be the next slice's first: it is read, not scanned, there)
        const auto t_scan0 = std::chrono::steady_clock::now();
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t nsl = std::max<size_t>(1, std::min<size_t>({(size_t)hw, (size_t)16, scan_len / ((size_t)1 << 20)}));
        std::vector<std::vector<size_t>> found(nsl);
        auto scan_slice = [&](size_t k) {
            const uint8_t* const lim = scan + scan_len - 1;   // (the last byte cannot start a marker)
            const uint8_t* q = scan + scan_len * k / nsl;
            const uint8_t* const end = std::min(lim, scan + scan_len * (k + 1) / nsl);
            std::vector<size_t>& out = found[k];
            out.reserve((size_t)(nint / nsl + 16));
            while (q < end) {
                q = static_cast<const uint8_t*>(std::memchr(q, 0xFF, (size_t)(end - q)));
                if (!q) break;
                if (q[1] >= 0xD0 && q[1] <= 0xD7) out.push_back((size_t)(q - scan));
                ++q;
            }
        };
        if (nsl == 1) {
            scan_slice(0);
        } else {
            std::vector<std::thread> th;
            for (size_t k = 1; k < nsl; ++k) th.emplace_back(scan_slice, k);
            scan_slice(0);
            for (auto& t : th) t.join();
        }
        std::vector<size_t> rst;
        rst.reserve((size_t)nint);
        for (const auto& v : found) rst.insert(rst.end(), v.begin(), v.end());
        if (std::getenv("KPEG_DEBUG"))
            std::fprintf(stderr, "kpeg_hip: restart-marker scan of %zu bytes: %zu markers, %zu host threads, %.3f ms\n", scan_len, rst.size(), nsl,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_scan0).count());
        if (rst.size() + 1 != nint) {
            root->last_error = "restart markers do not match the restart interval";
            return KPEG_HIP_E_STREAM;
        }
        for (int g = 0; g < ngpu; ++g) {
            Stripe& s = st[(size_t)g];
            s.r0 = std::min((uint32_t)g * rows_per, mh);
            s.nr = std::min(rows_per, mh - s.r0);
            if (!s.nr) continue;
            if (((uint64_t)s.r0 * mw) % f->restart_interval) {
                root->last_error = "stripe boundaries must coincide with restart intervals";
                return KPEG_HIP_E_ARG;
            }
            const uint64_t i0 = (uint64_t)s.r0 * mw / f->restart_interval;
            const uint64_t i1 = ((uint64_t)(s.r0 + s.nr) * mw + f->restart_interval - 1) / f->restart_interval;
            s.b0 = i0 == 0 ? 0 : rst[i0 - 1] + 2;
            s.b1 = i1 >= nint ? scan_len : rst[i1 - 1];
        }
    }
    if (root_is_device) {
        // stripes travel GPU to GPU (hipMemcpyPeerAsync): direct access over xGMI is switched on once per pair; a pair without it is
        // said so (the copy would be staged through the host: correct, and an order of magnitude slower)
        for (int g = 1; g < ngpu; ++g) {
            const int src = ctxs[g]->device, dst = root->device;
            if (src == dst) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, src, dst) != hipSuccess || !can) {
                root->last_error = "GPU " + std::to_string(src) + " has no peer access to GPU " + std::to_string(dst) + " (the stripes would be staged through host memory)";
                return KPEG_HIP_E_UNSUPPORTED;
            }
            if (hipSetDevice(src) != hipSuccess) return KPEG_HIP_E_DEVICE;
            const hipError_t pe = hipDeviceEnablePeerAccess(dst, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
                root->last_error = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe);
                (void)hipSetDevice(root->device);
                return KPEG_HIP_E_DEVICE;
            }
            (void)hipGetLastError();   // (already enabled: not an error to carry along)
        }
        (void)hipSetDevice(root->device);
    }
    const size_t pitch = (size_t)f->width * 3;
    // One host thread per GPU (a context is used by one thread at a time; different contexts are independent): upload,
    // decode and the stripe's way to the root are enqueued on that GPU's stream and waited for there, so no GPU waits
    // for the host to have finished enqueueing another GPU's work.
    std::vector<int> err((size_t)ngpu, KPEG_HIP_OK);
    auto work = [&](int g) {
        kpeg_hip_ctx* c = ctxs[g];
        const Stripe& s = st[(size_t)g];
        int& e = err[(size_t)g];
        if (!s.nr) return;
        if (hipSetDevice(c->device) != hipSuccess) { e = KPEG_HIP_E_DEVICE; return; }
        const size_t len = s.b1 - s.b0, sbytes = (size_t)s.nr * 8 * pitch;
        uint8_t* const home = rgb_root + (size_t)s.r0 * 8 * pitch;   // where the stripe belongs
        if ((e = grow(c, &c->d_scan, &c->scan_cap, len + 64))) return;
        uint8_t* dst = home;
        if (!(root_is_device && c->device == root->device)) {
            if ((e = grow(c, &c->d_rgb, &c->rgb_cap, sbytes))) return;
            c->rgb_gen++;
            dst = (uint8_t*)c->d_rgb;
        }
        if (hipMemcpyAsync(c->d_scan, scan + s.b0, len, hipMemcpyHostToDevice, c->stream) != hipSuccess) { e = KPEG_HIP_E_DEVICE; return; }
        if ((e = kpeg_hip_decode_stripe_dev(c, f, (const uint8_t*)c->d_scan, len, s.r0, s.nr, dst))) return;
        if (dst != home) {
            const hipError_t he = root_is_device
                ? hipMemcpyPeerAsync(home, root->device, dst, c->device, sbytes, c->stream)   // over xGMI, behind this stripe's decode
                : hipMemcpyAsync(home, dst, sbytes, hipMemcpyDeviceToHost, c->stream);        // this GPU's own PCIe link
            if (he != hipSuccess) { e = KPEG_HIP_E_DEVICE; return; }
        }
        e = kpeg_hip_sync(c);
    };
    {
        // contexts that share a GPU-side object (the same context listed twice) must not run concurrently: one thread each
        // only for distinct contexts
        std::vector<std::thread> th;
        for (int g = 1; g < ngpu; ++g) {
            bool dup = false;
            for (int k = 0; k < g; ++k) dup = dup || ctxs[k] == ctxs[g];
            if (dup) work(g);
            else th.emplace_back(work, g);
        }
        work(0);
        for (auto& t : th) t.join();
    }
    (void)hipSetDevice(root->device);
    for (int g = 0; g < ngpu; ++g)
        if (err[(size_t)g]) {
            const std::string why = ctxs[g]->last_error;   // (may be root's own: copy before it is overwritten)
            root->last_error = "stripe " + std::to_string(g) + ": " + std::string(kpeg_hip_strerror(err[(size_t)g])) + " (" + why + ")";
            return err[(size_t)g];
        }
    return KPEG_HIP_OK;
}

extern "C" int kpeg_hip_decode_sharded(kpeg_hip_ctx* const* ctxs, int ngpu, const kpeg_frame* f, const uint8_t* scan, size_t scan_len,
                                       uint8_t* rgb_root)
{
    return decode_sharded_impl(ctxs, ngpu, f, scan, scan_len, rgb_root, false);
}

extern "C" int kpeg_hip_decode_sharded_dev(kpeg_hip_ctx* const* ctxs, int ngpu, const kpeg_frame* f, const uint8_t* scan, size_t scan_len,
                                           uint8_t* d_rgb_root)
{
    return decode_sharded_impl(ctxs, ngpu, f, scan, scan_len, d_rgb_root, true);
}

// test hook: key 1 = number of sync passes enqueued (0 = default), key 2 = K1's warm-up sub-sequences (< 0 = default),
// key 3 = images per fused-batch chunk (0 = default), key 4 = sub-sequence size (0 = from the bit rate and the picture's size, else forced: 64 = the small
// pictures', >= 384 the dense one, anything else the sparse one), key 5 = bound of the device-side waits between workgroups in microseconds (0 = defaults), key 6 = fault
// injection: bit 0 K0's, bit 1 the chained K1 pass's first workgroup never publishes (its successors must time out), bit 2 K0 publishes no
// aggregates, bit 3 every third workgroup of k_sync_write gives the call up (the kernel's second launch then decodes it), key 7 = coefficient
// layout between K2 and K4: 0 = chosen per call, 1 = always dense, 2 = the compact stream wherever it is possible, key 8 = K0 even
// where K1 and K2 could un-stuff for themselves, key 9 = k_sync_write (K1's pass 0 and K2 in one kernel) where it applies: 1 (default) / 0
extern "C" int kpeg_hip_debug_set(kpeg_hip_ctx* ctx, int key, int value)
{
    if (!ctx) return KPEG_HIP_E_ARG;
    if (key == 1) ctx->sync_passes = value;
    else if (key == 2) ctx->warm = value;
    else if (key == 3) ctx->batch_chunk = value > 0 && value <= 4096 ? value : 4096;
    else if (key == 4) ctx->subseq = value;
    else if (key == 5) ctx->spin_ticks = value > 0 ? (unsigned long long)value * 100ull : 0ull;   // microseconds
    else if (key == 6) ctx->fault = (uint32_t)value;
    else if (key == 7) ctx->coef_layout = value;
    else if (key == 8) ctx->force_k0 = value ? 1u : 0u;   // K0 even for one image without restart markers (K1 and K2 un-stuff for themselves there)
    else if (key == 9) ctx->fused_slots = value ? ctx->fused_slots_dev : 0u;   // k_sync_write (K1's pass 0 and K2 in one kernel) where it applies
    else return KPEG_HIP_E_ARG;
    return KPEG_HIP_OK;
}

// test hook: raw status words of the last synchronised call
extern "C" int kpeg_hip_debug_words(kpeg_hip_ctx* ctx, uint32_t* out, int n)
{
    if (!ctx || !out) return KPEG_HIP_E_ARG;
    for (int i = 0; i < n && i < (int)STATUS_WORDS; ++i) out[i] = ctx->status_seen[i];
    return KPEG_HIP_OK;
}

// test hook, host only (no device needed): the first-level Huffman tables as the kernels get them -- lut[4][512] (one symbol
// per entry, [class * 2 + id]) and lutx[4][512] (K1's two-symbol entries, same order) -- for tests/test_tables.py, which checks every
// two-symbol entry against two steps through the one-symbol table.  lut_bits receives LUT_BITS.
extern "C" int kpeg_hip_debug_entropy_luts(const kpeg_frame* frame, uint32_t* lut, uint32_t* lutx, int* lut_bits)
{
    if (!frame || !lut || !lutx || !lut_bits) return KPEG_HIP_E_ARG;
    static kpeg_dev::EntropyTables t;   // (43 KB: not on the stack)
    if (kpeg_dev::build_entropy_tables(frame, &t) != 0) return KPEG_HIP_E_ARG;
    std::memcpy(lut, t.lut, sizeof(t.lut));
    std::memcpy(lutx, t.lutx, sizeof(t.lutx));
    *lut_bits = kpeg_dev::LUT_BITS;
    return KPEG_HIP_OK;
}

#if KPEG_SYNC_STATS
// experiment builds only: per-wavefront timelines of K1 pass 0 (which = 0) and K2 (1), tools/sync_dbg.py
extern "C" int kpeg_hip_debug_entropy_stamps(int which, unsigned long long* out, int n)
{
    if (which < 0 || which > 1 || n > 8192 * 16) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kpeg_dev::g_ent_stamp), (size_t)n * 8, (size_t)which * 8192 * 16 * 8) == hipSuccess ? 0 : -2;
}
#endif
#ifdef KPEG_K4_STAMP
// diagnostic build only (tools/k4_clock.py): per-wavefront start/end stamps of the last K4 launch
extern "C" int kpeg_hip_debug_k4_stamps(unsigned long long* out, int n)
{
    if (n > 8192 * 8) n = 8192 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kpeg_dev::g_k4_stamp), (size_t)n * 8) == hipSuccess ? 0 : -2;
}
#endif
