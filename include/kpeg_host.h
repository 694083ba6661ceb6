/* include/kpeg_host.h -- small C shim over the C++ host API (kpeg::JPEGDecoder), for callers
 * that cannot include C++ headers (ctypes in tests/ and bench.py).  Implemented in libkpeg.so.
 *
 * kpeg_host_parse runs the product's marker parser (the same code path kpeg::JPEGDecoder::
 * decodeImageFile takes up to the seam, src/Decoder.cpp:105-133 in the reference) on an
 * in-memory file and returns what is handed to the GPU path.
 */
#ifndef KPEG_HOST_H
#define KPEG_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "kpeg_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KPEG_PARSE_ALLOW_DRI 1u /* extension: accept DRI/RSTn (the reference rejects them) */
#define KPEG_PARSE_ALLOW_GRAY 2u /* extension: accept one-component (grayscale) baseline files (the reference fails on them) */
#define KPEG_PARSE_ALLOW_420 8u /* extension: accept 4:2:0 files, any size (the reference answers TERMINATE on subsampled files) */
#define KPEG_PARSE_ALLOW_ANY_SIZE 4u /* extension: accept widths / heights that are not multiples of 8 (the reference reads past its MCU vector on them) */

/* Returns the reference's JPEGDecoder::ResultCode (0 SUCCESS, 1 TERMINATE, 2 ERROR,
 * 3 DECODE_INCOMPLETE, 4 DECODE_DONE), or -1 when the tables are outside the supported
 * layout.  On DECODE_DONE: *frame is filled, the entropy-coded segment is copied to `scan`
 * (capacity scan_cap bytes; size + 1 always suffices) and its length stored in *scan_len. */
int kpeg_host_parse(const uint8_t* file, size_t size, unsigned flags, kpeg_frame* frame, uint8_t* scan, size_t scan_cap,
                    size_t* scan_len);

/* Whole path for one file on disk: parse, GPU decode, write <name>.ppm next to it (the CLI's
 * behaviour, reference main.cpp:19-33).  Returns the ResultCode. */
int kpeg_host_decode_file(const char* path, unsigned flags);

/* Byte offsets of the restart markers inside a still-stuffed scan: offsets[i] = position of the
 * FF of the i-th RSTn.  Returns the number of markers found (may exceed cap; only cap are stored). */
size_t kpeg_host_restart_offsets(const uint8_t* scan, size_t n, uint64_t* offsets, size_t cap);

/* kpeg::HuffmanTree::contains() on a tree built from (counts, symbols): writes the reference's
 * answer ("", "EOB" or the decimal symbol) into out.  For the known-answer tests. */
int kpeg_host_huffman_contains(const uint8_t counts[16], const uint8_t* symbols, const char* bits, char* out, size_t cap);

/* kpeg::bitStringtoValue / valueToBitString / getValueCategory (src/Image.cpp:258-320). */
int kpeg_host_bitstring_to_value(const char* bits);
int kpeg_host_value_to_bitstring(int value, char* out, size_t cap);

/* Hash of the host sources this library was built from (libkpeg_amd/build.py rebuilds on a mismatch). */
const char* kpeg_host_build_hash(void);

/* kpeg::isValidFilename (include/Utility.hpp:16-38). */
int kpeg_host_is_valid_filename(const char* name);

#ifdef __cplusplus
}
#endif
#endif
