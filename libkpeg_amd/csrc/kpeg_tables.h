// libkpeg_amd/csrc/kpeg_tables.h -- constant tables shared by host and device code.
#pragma once
#include <stdint.h>

// Error flags (device status word [1]): 1 restart markers do not match the restart interval, 2 more sub-sequences than
// planned for, 8 no such Huffman code, 16 DC symbol with a run nibble, 32 run past the end of a block, 64 / 128 a segment's
// last block is incomplete / missing, 256 a wait on another workgroup timed out (kpeg_hip_sync: KPEG_HIP_E_DEVICE),
// 512 an accumulated DC value outside int16 (the reference keeps ints, MCU.cpp:107-112: outside the contract).
#define KPEG_ERR_TIMEOUT 256u
#define KPEG_ERR_DC_RANGE 512u

// cos((2a+1)*b*M_PI/16.0) as evaluated by glibc's libm on x86-64 (the values the
// reference's MCU::computeIDCT sees, src/MCU.cpp:192-193), row a, column b, written
// as hex doubles so that no libm is involved on the GPU box.  tests/test_tables.py
// checks them against the run-time libm and against tests/golden/cos_table.hex.
static const double KPEG_COS_TABLE[64] = {
    0x1.0000000000000p+0, 0x1.f6297cff75cb0p-1, 0x1.d906bcf328d46p-1, 0x1.a9b66290ea1a3p-1, 0x1.6a09e667f3bcdp-1, 0x1.1c73b39ae68c9p-1, 0x1.87de2a6aea964p-2, 0x1.8f8b83c69a60dp-3,
    0x1.0000000000000p+0, 0x1.a9b66290ea1a3p-1, 0x1.87de2a6aea964p-2, -0x1.8f8b83c69a608p-3, -0x1.6a09e667f3bccp-1, -0x1.f6297cff75cb0p-1, -0x1.d906bcf328d47p-1, -0x1.1c73b39ae68c8p-1,
    0x1.0000000000000p+0, 0x1.1c73b39ae68c9p-1, -0x1.87de2a6aea962p-2, -0x1.f6297cff75cb0p-1, -0x1.6a09e667f3bcep-1, 0x1.8f8b83c69a60cp-3, 0x1.d906bcf328d44p-1, 0x1.a9b66290ea1a5p-1,
    0x1.0000000000000p+0, 0x1.8f8b83c69a60dp-3, -0x1.d906bcf328d46p-1, -0x1.1c73b39ae68c8p-1, 0x1.6a09e667f3bcbp-1, 0x1.a9b66290ea1a5p-1, -0x1.87de2a6aea965p-2, -0x1.f6297cff75cb2p-1,
    0x1.0000000000000p+0, -0x1.8f8b83c69a608p-3, -0x1.d906bcf328d47p-1, 0x1.1c73b39ae68c5p-1, 0x1.6a09e667f3bcep-1, -0x1.a9b66290ea1a2p-1, -0x1.87de2a6aea971p-2, 0x1.f6297cff75cb0p-1,
    0x1.0000000000000p+0, -0x1.1c73b39ae68c6p-1, -0x1.87de2a6aea96dp-2, 0x1.f6297cff75cb0p-1, -0x1.6a09e667f3bc5p-1, -0x1.8f8b83c69a602p-3, 0x1.d906bcf328d46p-1, -0x1.a9b66290ea1a1p-1,
    0x1.0000000000000p+0, -0x1.a9b66290ea1a4p-1, 0x1.87de2a6aea967p-2, 0x1.8f8b83c69a61dp-3, -0x1.6a09e667f3bc9p-1, 0x1.f6297cff75cb2p-1, -0x1.d906bcf328d43p-1, 0x1.1c73b39ae68c2p-1,
    0x1.0000000000000p+0, -0x1.f6297cff75cb0p-1, 0x1.d906bcf328d44p-1, -0x1.a9b66290ea1a2p-1, 0x1.6a09e667f3bc4p-1, -0x1.1c73b39ae68c2p-1, 0x1.87de2a6aea95fp-2, -0x1.8f8b83c69a616p-3,
};

// zig-zag index -> row*8+col (zzOrderToMatIndices, src/Transform.cpp:5-27)
static const uint8_t KPEG_ZZ_TO_NATURAL[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};
