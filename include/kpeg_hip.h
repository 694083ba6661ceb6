/* include/kpeg_hip.h -- C ABI of the MI355X (gfx950) decode path.
 *
 * This is the drop-in boundary for libKPEG's per-MCU hot loop.  The reference has
 * no plugin/FFI interface; the seam these entry points replace is the pair of calls
 *
 *     decodeScanData();                          // src/Decoder.cpp:137  (:655-855)
 *     m_image.createImageFromMCUs( m_MCU );      // src/Decoder.cpp:138  (src/Image.cpp:20-86)
 *
 * inside JPEGDecoder::decodeImageFile (src/Decoder.cpp:135-139), after the marker loop
 * (:105-133) has filled m_QTables (include/Decoder.hpp:110), m_huffmanTable[2][2] (:117),
 * the image dimensions and m_scanData (:125).  Everything below that seam --
 * byteStuffScanData (:621-653), the Huffman bit loop (:694-803), HuffmanTree::contains
 * (src/HuffmanTree.cpp:164-193), bitStringtoValue (src/Image.cpp:285-302),
 * MCU::constructMCU / computeIDCT / performLevelShift / convertYCbCrToRGB
 * (src/MCU.cpp:64-279) and the MCU tiling of Image::createImageFromMCUs -- runs on the
 * GPU.  Output pixels are bit-identical to the reference's.
 *
 * Conventions: plain C, no exceptions across the boundary, 0 = success, negative =
 * error (kpeg_hip_strerror).  The caller owns every buffer it passes.  A context is
 * bound to one device and one stream and must be used from one host thread at a time.
 * There is no CPU fallback: every entry point fails with KPEG_HIP_E_DEVICE if the GPU
 * path cannot run.
 */
#ifndef KPEG_HIP_H
#define KPEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KPEG_HIP_ABI_VERSION 2

enum {
    KPEG_HIP_OK = 0,
    KPEG_HIP_E_ARG = -1,      /* null pointer, zero size, dimensions not multiples of 8 ...  */
    KPEG_HIP_E_DEVICE = -2,   /* HIP runtime error / no gfx950 device (message: last_error)  */
    KPEG_HIP_E_TABLES = -3,   /* Huffman table is not a valid prefix code                    */
    KPEG_HIP_E_STREAM = -4,   /* entropy-coded data ended early or held an invalid code      */
    KPEG_HIP_E_NOMEM = -5,
    KPEG_HIP_E_UNSUPPORTED = -6
};

typedef struct kpeg_hip_ctx kpeg_hip_ctx;

/* kpeg::HuffmanTable (include/Types.hpp:116) flattened: counts[i] codes of length i+1,
 * symbols in code order. */
#define KPEG_FRAME_420 0x0203u /* kpeg_frame.components: three components, luma sampled 2x2 */

typedef struct kpeg_dht {
    uint8_t counts[16];
    uint8_t symbols[256];
} kpeg_dht;

/* What the marker parser knows when it reaches the seam. */
typedef struct kpeg_frame {
    uint32_t width, height;     /* SOF0 dimensions (Decoder.cpp:361); multiples of 8 -- any
                                   size 1..65535 at the whole-image entry points, see below   */
    uint16_t qt[2][64];         /* m_QTables[0], [1]: zig-zag order as stored (Decoder.cpp:278)
                                   [0] -> Y, [1] -> Cb and Cr (hard-wired, MCU.cpp:110)       */
    kpeg_dht dht[2][2];         /* m_huffmanTable[class 0=DC,1=AC][id]; id 0 -> Y,
                                   id 1 -> Cb, Cr (hard-wired, Decoder.cpp:704)               */
    uint32_t restart_interval;  /* MCUs per restart interval; 0 = a stream the reference
                                   accepts (it rejects DRI)                    */
    uint32_t components;        /* 0 or 3 = Y Cb Cr 4:4:4, what the reference decodes; 1 = grayscale (extension, off by
                                   default in the host parser: the reference reads three component triples whatever
                                   SOF0 says, src/Decoder.cpp:339, and fails on such files): one block per MCU decoded with
                                   table id 0 through the same per-block arithmetic, R = G = B = clamp(Y);
                                   KPEG_FRAME_420 = Y Cb Cr 4:2:0 (extension, off by default in the host parser: the
                                   reference answers TERMINATE on sampling factors other than 1x1): 16x16 MCUs of six
                                   blocks through the same per-block arithmetic, every chroma sample repeated 2x2;
                                   whole-image and batch entry points, any width / height                               */
} kpeg_frame;

/* Per-call device timings (milliseconds, HIP events on the context's stream between the kernels, so a span is a kernel
 * plus the launch gap in front of it), valid after kpeg_hip_sync() when profiling is enabled.  0 for kernels that did
 * not run.  What the spans bracket depends on the path the call took:
 *   separate launches (restart segments, batches, dense streams, KPEG_FUSED=0): the names say it;
 *   one image on the compact stream (k_sync_write: K1's pass 0 and K2 in one kernel): huff_sync_ms is that kernel together
 *   with its second, strict launch, which leaves at once unless the first gave the call up; huff_scan_ms, huff_write_ms and
 *   dc_ms are the gaps between event records (no kernel runs in them). */
typedef struct kpeg_hip_timings {
    float unstuff_ms;      /* K0: FF00 removal / restart-segment scan (restart segments and batches only: one image
                              without restart markers is un-stuffed by K1 and K2 as they stage it)                       */
    float huff_sync_ms;    /* K1: self-synchronising sub-sequence decode, all its launches; or k_sync_write's two launches
                              (see above)                                                                                 */
    float huff_scan_ms;    /*     (the scan of the totals runs inside K1's last launch: this span is a launch gap)       */
    float huff_write_ms;   /* K2: coefficient-writing decode pass (k_write); a gap behind k_sync_write                   */
    float dc_ms;           /*     (no DC kernel any more: the DC sums ride in K1's totals; a launch gap)                 */
    float idct_ms;         /* K4: dequantise + IDCT + level shift + colour + tiled RGB store                             */
    float total_ms;        /* sum of the spans, first event -> last event                                                */
    uint32_t sync_rounds;  /* launches of K1 that had work: 1 = pass 0 (or k_sync_write) settled every workgroup,
                              2 = the verifying launch (or k_sync_write's second launch) decoded again, 3 = the chained
                              launch rippled                                                                              */
    uint32_t exact_pixels; /* K4: pixels that took the reference-order re-evaluation                                     */
} kpeg_hip_timings;

/* ---- lifecycle ------------------------------------------------------------------------- */
int kpeg_hip_abi_version(void);
/* Hash of the kernel sources and compiler flags this library was built from; libkpeg_amd/build.py rebuilds the
 * library when it differs from the hash of the sources in the tree (a stale binary must not reach the GPU). */
const char* kpeg_hip_build_hash(void);
int kpeg_hip_create(kpeg_hip_ctx** ctx, int device);
void kpeg_hip_destroy(kpeg_hip_ctx* ctx);
const char* kpeg_hip_strerror(int code);
const char* kpeg_hip_last_error(const kpeg_hip_ctx* ctx);
/* Launch on a caller-owned hipStream_t (created with hipStreamCreate*; e.g. a PyTorch side stream); NULL = the
 * context's own non-blocking stream.  NOTE: the legacy default stream's handle IS NULL (PyTorch's default stream reports
 * 0), so it cannot be selected here: a caller whose buffers are produced on the default stream either synchronises before
 * the call or does that work on a created stream and passes it.  Work queued on the stream in use before the switch is
 * waited for by the new stream (the scratch buffers are shared). */
int kpeg_hip_set_stream(kpeg_hip_ctx* ctx, void* hip_stream);
/* Wait for the context's stream and return the deferred status of the last *_dev call
 * (kernels report corrupt entropy data through a device-side flag). */
int kpeg_hip_sync(kpeg_hip_ctx* ctx);
int kpeg_hip_set_profiling(kpeg_hip_ctx* ctx, int enable);
int kpeg_hip_get_timings(kpeg_hip_ctx* ctx, kpeg_hip_timings* out);

/* ---- host-buffer entry points (synchronous; H2D and D2H inside) ------------------------ */

/* Replaces MCU::constructMCU's dequantise/de-zig-zag + computeIDCT + performLevelShift +
 * convertYCbCrToRGB (MCU.cpp:110-279) + Image::createImageFromMCUs (Image.cpp:20-86) for a
 * whole image whose Huffman decode was done by the caller.
 *   coef: [mcu][comp 0..2][row u][col v] int16 -- the quantised (NOT dequantised) content
 *         of MCU::m_8x8block just before MCU.cpp:110, i.e. natural order, absolute DC
 *         (MCU.cpp:107-108), quirk Q1 applied (MCU.cpp:97-100).  mcu = tile row-major.
 *   rgb : height*width*3 bytes, row-major R,G,B -- the bytes Image::dumpRawData writes
 *         after its header (Image.cpp:129-135). */
int kpeg_hip_idct_colour(kpeg_hip_ctx* ctx, const kpeg_frame* frame, const int16_t* coef, uint8_t* rgb);

/* Replaces decodeScanData() + createImageFromMCUs().
 *   scan: the entropy-coded segment exactly as scanImageData collects it
 *         (Decoder.cpp:544-574): every byte after the SOS header up to, not including,
 *         the FF D9; still byte-stuffed. */
int kpeg_hip_decode_scan(kpeg_hip_ctx* ctx, const kpeg_frame* frame, const uint8_t* scan, size_t scan_len,
                         uint8_t* rgb);
/* Extension (off by default in the host parser, KPEG_PARSE_ALLOW_ANY_SIZE): kpeg_hip_decode_scan, _scan_dev, _scan_resident,
 * kpeg_hip_download_bands, kpeg_hip_decode_stripe_dev (MCU rows of the padded picture; a stripe's last pixel rows are the
 * picture's) and kpeg_hip_decode_batch, _batch_dev also take widths / heights that are not multiples of 8.  All ceil(w/8) * ceil(h/8) MCUs of
 * the padded picture are decoded and the rows and columns Image::createImageFromMCUs pops (Image.cpp:26-27,73-84) are
 * cropped on the device; rgb is height*width*3 bytes as ever.  (The reference itself decodes (w*h)/64 MCUs and tiles
 * more than it has: undefined behaviour, so nothing to be bit-identical with but its output for the same scan at the
 * padded size -- which is what the tests pin.)  The kernels' own entry points (kpeg_hip_entropy_decode_dev, kpeg_hip_idct_colour*)
 * and the sharded ones work on whole blocks and keep the multiple-of-8 contract. */

/* The number of calls that have written pixels into the context's resident buffer so far.  A caller that leaves a decoded image
 * there (kpeg_hip_decode_scan_resident) notes it and checks it before kpeg_hip_download_bands: any later decode, batch or
 * sharded call on the context changes it, and the pixels are then another picture's. */
unsigned long long kpeg_hip_resident_generation(const kpeg_hip_ctx* ctx);
/* The same decode with the pixels left on the device, in a buffer the context owns (valid until the context's next
 * decode), and their download in row bands through two pinned bounce buffers: while band k+1 crosses PCIe, `sink` is
 * called with band k (from the calling thread), e.g. to fwrite it -- Image::dumpRawData's 99.5 MB (8K) then reach the
 * file without ever being assembled in pageable memory (reference sink: src/Image.cpp:108-140).  band_rows = 0 picks
 * bands of about 8 MiB.  sink returns non-zero to abort (KPEG_HIP_E_ARG is returned). */
typedef int (*kpeg_hip_band_sink)(void* user, uint32_t first_row, uint32_t rows, const uint8_t* rgb, size_t bytes);
int kpeg_hip_decode_scan_resident(kpeg_hip_ctx* ctx, const kpeg_frame* frame, const uint8_t* scan, size_t scan_len);
int kpeg_hip_download_bands(kpeg_hip_ctx* ctx, const kpeg_frame* frame, uint32_t band_rows, kpeg_hip_band_sink sink, void* user);

/* `count` independent images of identical geometry and tables (throughput mode, BASELINE config 4; the
 * reference's counterpart is a loop over JPEGDecoder::decodeImageFile, src/main.cpp:19-33).  Chunks of
 * images alternate between two internal lanes (stream + buffers + pinned staging each): a chunk's scans go
 * up in one copy and are decoded by the fused batch path (see kpeg_hip_decode_batch_dev) while the
 * previous chunk is downloaded into rgbs[].  Returns after the last pixel is in rgbs[]; a stream error of
 * any image fails the call (kpeg_hip_last_error names the chunk). */
int kpeg_hip_decode_batch(kpeg_hip_ctx* ctx, int count, const kpeg_frame* frame, const uint8_t* const* scans,
                          const size_t* scan_lens, uint8_t* const* rgbs);

/* Row-stripe sharding of ONE image over several GPUs from a single host thread (BASELINE config 5; the reference
 * has no counterpart: it rejects DRI, src/Decoder.cpp:58-74, and is single-threaded).  ctxs[g] is a context created
 * on the g-th GPU to use (two contexts on one GPU also work); frame->restart_interval must be non-zero and whole
 * MCU rows must start on restart-interval boundaries.  `scan` is the whole entropy-coded segment in host memory: it
 * is cut at its RSTn markers, stripe g (MCU rows [g*R, (g+1)*R), R = ceil(rows / ngpu)) is uploaded to ctxs[g]'s
 * GPU and decoded there with kpeg_hip_decode_stripe_dev, all GPUs concurrently.
 *   kpeg_hip_decode_sharded      rgb_root = host buffer of height*width*3 bytes: every GPU downloads its own rows
 *                                straight into it over its own PCIe link (no gather through one GPU).
 *   kpeg_hip_decode_sharded_dev  d_rgb_root = device buffer of height*width*3 bytes on ctxs[0]'s GPU: the other
 *                                GPUs' stripes arrive there by peer copies over xGMI (hipMemcpyPeerAsync), each as
 *                                soon as its stripe is decoded.
 * Both return after everything has arrived; the first error of any stripe is returned (kpeg_hip_last_error(ctxs[0])
 * names the stripe). */
int kpeg_hip_decode_sharded(kpeg_hip_ctx* const* ctxs, int ngpu, const kpeg_frame* frame, const uint8_t* scan, size_t scan_len,
                            uint8_t* rgb_root);
int kpeg_hip_decode_sharded_dev(kpeg_hip_ctx* const* ctxs, int ngpu, const kpeg_frame* frame, const uint8_t* scan, size_t scan_len,
                                uint8_t* d_rgb_root);

/* ---- device-resident entry points (asynchronous on the context's stream) --------------- */
/* All pointers are device pointers.  Errors found by kernels surface at kpeg_hip_sync(). */
int kpeg_hip_idct_colour_dev(kpeg_hip_ctx* ctx, const kpeg_frame* frame, const int16_t* d_coef, uint8_t* d_rgb);
int kpeg_hip_decode_scan_dev(kpeg_hip_ctx* ctx, const kpeg_frame* frame, const uint8_t* d_scan, size_t scan_len,
                             uint8_t* d_rgb);
/* Row-stripe sharding (frame->restart_interval must divide width/8 * k): decode only the
 * MCU rows [first_mcu_row, first_mcu_row + mcu_rows) of the image.  d_scan holds the bytes
 * of exactly those restart intervals (RSTn markers between them included); d_rgb receives
 * mcu_rows*8 pixel rows.  One rank per GPU calls this on its stripe; the caller gathers. */
int kpeg_hip_decode_stripe_dev(kpeg_hip_ctx* ctx, const kpeg_frame* frame, const uint8_t* d_scan, size_t scan_len,
                               uint32_t first_mcu_row, uint32_t mcu_rows, uint8_t* d_rgb);
/* kpeg_hip_decode_batch with device-resident scans and outputs (arrays of device pointers held on the
 * host; rgb buffers 16-byte aligned): the images are decoded as the restart segments of one virtual
 * stream, i.e. by ONE set of kernel launches per chunk of up to 4096 images / 256 MiB of scan data.
 * Asynchronous on the context's stream for a single chunk; errors surface at kpeg_hip_sync(). */
int kpeg_hip_decode_batch_dev(kpeg_hip_ctx* ctx, int count, const kpeg_frame* frame, const uint8_t* const* d_scans,
                              const size_t* scan_lens, uint8_t* const* d_rgbs);
/* Entropy decode only: d_coef receives the coefficient layout of kpeg_hip_idct_colour. */
int kpeg_hip_entropy_decode_dev(kpeg_hip_ctx* ctx, const kpeg_frame* frame, const uint8_t* d_scan, size_t scan_len,
                                int16_t* d_coef);

/* ---- knobs for tests and benchmarks ------------------------------------------------------- */
/* IDCT kernel variant: 0 = fast path with exact re-evaluation of unsafe samples (default),
 * 1 = reference-order evaluation of every sample (slow, used as an on-device cross-check),
 * 2 = fast path WITHOUT the re-evaluation: wrong pixels, timing experiments only. */
int kpeg_hip_set_idct_mode(kpeg_hip_ctx* ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif /* KPEG_HIP_H */
