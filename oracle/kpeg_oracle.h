/* oracle/kpeg_oracle.h -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement (plain C) of libKPEG's decode-to-PPM path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it, and only
 * as the checker.  Every function cites the reference lines it follows.
 *
 * Pinning: the restatement is checked bit-for-bit against the real reference
 * binary (oracle/_ref/kpeg_ref, built by oracle/Makefile from /root/reference)
 * on misc/images/lena.jpg (SHA-256 in tests/golden/) and on seeded synthetic and
 * Pillow-encoded JPEGs; see tests/test_oracle.py and tests/golden/make_golden.py.
 */
#ifndef KPEG_ORACLE_H
#define KPEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* JPEGDecoder::ResultCode, include/Decoder.hpp:29-36 (same numeric values). */
enum {
    KPEG_ORACLE_SUCCESS = 0,
    KPEG_ORACLE_TERMINATE = 1,
    KPEG_ORACLE_ERROR = 2,
    KPEG_ORACLE_DECODE_INCOMPLETE = 3,
    KPEG_ORACLE_DECODE_DONE = 4,
    /* not a reference code: the input leaves the contract of SURVEY.md A.1
     * (the reference would hit undefined behaviour or throw). */
    KPEG_ORACLE_OUT_OF_CONTRACT = 100
};

typedef struct {
    uint8_t counts[16];   /* number of codes of length 1..16 */
    uint8_t symbols[256]; /* symbols in code order           */
    int nsymbols;
    int defined;
} kpeg_oracle_dht;

typedef struct {
    uint32_t width, height;
    int nqt;                  /* number of DQT tables pushed (Decoder.cpp:264) */
    uint16_t qt[4][64];       /* zig-zag order, as stored by the reference     */
    kpeg_oracle_dht dht[2][2];/* [class 0=DC,1=AC][id 0/1]                     */
    uint8_t* scan;            /* entropy-coded bytes as appended by
                                 scanImageData (Decoder.cpp:544-574); malloc'd */
    size_t scan_len;
    int saw_sos;
} kpeg_oracle_jfif;

/* Marker loop of JPEGDecoder::decodeImageFile (Decoder.cpp:105-133) with the
 * per-segment parsers (:164-530, :579-619).  Returns a ResultCode. */
int kpeg_oracle_parse(const uint8_t* file, size_t n, kpeg_oracle_jfif* out);
void kpeg_oracle_jfif_free(kpeg_oracle_jfif* j);

/* JPEGDecoder::byteStuffScanData (Decoder.cpp:621-653). out may alias in.
 * Returns the new length. */
size_t kpeg_oracle_unstuff(const uint8_t* in, size_t n, uint8_t* out);
/* literal erase-in-place form of the same loop (O(n^2)); out must not alias in */
size_t kpeg_oracle_unstuff_literal(const uint8_t* in, size_t n, uint8_t* out);

/* Bit loop of JPEGDecoder::decodeScanData (Decoder.cpp:655-855) followed by the
 * RLE walk / DC prediction of MCU::constructMCU (MCU.cpp:64-108) but NOT the
 * dequantisation: coef[mcu][comp][k] is the zig-zag-ordered quantised value
 * with absolute DC and quirk Q1 applied.  `bits` is the un-stuffed stream.
 * Returns 0, or KPEG_ORACLE_OUT_OF_CONTRACT (truncated / invalid stream).
 * If bits_used != NULL it receives the number of bits consumed. */
int kpeg_oracle_entropy_decode(const kpeg_oracle_jfif* j, const uint8_t* bits, size_t nbytes,
                               uint32_t nmcu, int16_t* coef, uint64_t* bits_used);

/* Same, but the stream consists of restart intervals of `interval` MCUs that are
 * separated by RSTn markers (FF D0..D7) in the still-stuffed scan `scan`; each
 * interval is byte-aligned and restarts the DC predictors at 0.  The reference
 * itself rejects DRI; this models "each interval re-wrapped as
 * its own JFIF and decoded by a fresh reference process" (SURVEY.md 8c). */
int kpeg_oracle_entropy_decode_rst(const kpeg_oracle_jfif* j, const uint8_t* scan, size_t nbytes,
                                   uint32_t nmcu, uint32_t interval, int16_t* coef);

/* MCU::constructMCU dequantisation + de-zig-zag (MCU.cpp:110-120, Transform.cpp:5-27),
 * MCU::computeIDCT (:172-216) for one component block.
 * zz: 64 quantised zig-zag coefficients, q: 64 zig-zag quantiser entries,
 * out: float icoeffs[x][y] (x = row). */
void kpeg_oracle_idct_block(const int16_t* zz, const uint16_t* q, float out[64]);

/* computeIDCT + performLevelShift (:218-245) + convertYCbCrToRGB (:247-279) +
 * Image::createImageFromMCUs tiling (Image.cpp:20-86).  rgb is H*W*3 bytes.
 * nthreads > 1 splits MCU rows over OpenMP threads (results are identical). */
void kpeg_oracle_idct_colour(const int16_t* coef, const uint16_t qt[2][64], uint32_t width,
                             uint32_t height, uint8_t* rgb, int nthreads);

/* Whole path.  On KPEG_ORACLE_DECODE_DONE *rgb is malloc'd (H*W*3). */
int kpeg_oracle_decode(const uint8_t* file, size_t n, uint8_t** rgb, uint32_t* width,
                       uint32_t* height, int nthreads);

/* Extension (parity unpinned: the reference cannot decode such files): a one-component baseline file through the
 * reference's per-block arithmetic, R = G = B = clamp(Y).  See kpeg_oracle.c. */
int kpeg_oracle_decode_gray(const uint8_t* file, size_t n, uint8_t** rgb, uint32_t* width, uint32_t* height, int nthreads);

/* Extension (parity unpinned: the reference's own tiling of such files reads past its MCU vector): any width / height,
 * all MCUs of the padded picture decoded, cropped as Image::createImageFromMCUs crops.  See kpeg_oracle.c. */
int kpeg_oracle_decode_any_size(const uint8_t* file, size_t n, uint8_t** rgb, uint32_t* width, uint32_t* height, int nthreads);

/* Extension (parity unpinned: the reference answers TERMINATE on sampling factors other than 1x1): 4:2:0 files, any size,
 * chroma samples repeated 2x2.  See kpeg_oracle.c. */
int kpeg_oracle_decode_420(const uint8_t* file, size_t n, uint8_t** rgb, uint32_t* width, uint32_t* height, int nthreads);

/* Header of Image::dumpRawData (Image.cpp:124-127). Returns its length. */
size_t kpeg_oracle_ppm_header(uint32_t width, uint32_t height, char* buf, size_t cap);

/* The 8x8 table cos((2a+1)*b*M_PI/16.0) (MCU.cpp:192-193) evaluated with this
 * host's libm at run time, a in 0..7 rows, b in 0..7 columns. */
void kpeg_oracle_cos_table(double out[64]);

/* zzOrderToMatIndices (Transform.cpp:5-27): zig-zag index -> row*8+col. */
int kpeg_oracle_zz_to_rowmajor(int k);

#ifdef __cplusplus
}
#endif
#endif
