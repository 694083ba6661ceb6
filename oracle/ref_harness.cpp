// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A small driver around the *real* reference decoder.  It is compiled by
// oracle/Makefile together with the reference's own translation units, taken
// where they lie under /root/reference (never copied into this repo), into
// oracle/_ref/kpeg_ref.  The binary is used
//   * to pin oracle/kpeg_oracle.c (the CPU restatement) bit-for-bit,
//   * to generate the fixtures under tests/golden/ (tests/golden/make_golden.py),
//   * optionally as bench.py's cpu_baseline leg (kind "reference").
//
// It only uses the reference's public API (kpeg::JPEGDecoder::open /
// decodeImageFile / dumpRawData, Decoder.hpp:40-60) plus, for the `stages`
// command, read access to private members through the usual test-only
// `#define private public` trick.
//
// Usage:
//   kpeg_ref decode <file.jpg>              -> writes <file>.ppm, prints JSON timing
//   kpeg_ref stages <file.jpg> <out.bin>    -> dumps per-MCU float IDCT output
//                                              (MCU::icoeffs, MCU.hpp:87) and the
//                                              final RGB blocks (m_8x8block)
//   kpeg_ref status <file.jpg>              -> prints the ResultCode only
//
// One process per image: the reference keeps its DC predictors in a static
// (MCU.cpp:53) that is never reset.

#include <array>
#include <bitset>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#define private public
#include "Decoder.hpp"
#include "Logger.hpp"
#undef private

static const char* codeName(kpeg::JPEGDecoder::ResultCode c)
{
    switch (c) {
        case kpeg::JPEGDecoder::SUCCESS: return "SUCCESS";
        case kpeg::JPEGDecoder::TERMINATE: return "TERMINATE";
        case kpeg::JPEGDecoder::ERROR: return "ERROR";
        case kpeg::JPEGDecoder::DECODE_INCOMPLETE: return "DECODE_INCOMPLETE";
        case kpeg::JPEGDecoder::DECODE_DONE: return "DECODE_DONE";
    }
    return "?";
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: kpeg_ref decode|stages|status <file.jpg> [out.bin]\n");
        return 2;
    }
    const std::string cmd = argv[1];
    const std::string file = argv[2];

    // The reference's logger must be given a stream and a level before first use
    // (Logger.hpp:95-96 are uninitialised otherwise).
    std::ofstream devnull("/dev/null");
    kpeg::Logger::get().setLogStream(devnull);
    kpeg::Logger::get().setLevel(kpeg::Logger::Level::ERROR);

    int rc = 0;
    try {
        kpeg::JPEGDecoder dec;
        if (!dec.open(file)) {
            std::printf("{\"status\": \"OPEN_FAILED\"}\n");
            return 1;
        }
        auto t0 = std::chrono::steady_clock::now();
        auto code = dec.decodeImageFile();
        auto t1 = std::chrono::steady_clock::now();
        double sec = std::chrono::duration<double>(t1 - t0).count();
        unsigned w = dec.m_image.getWidth(), h = dec.m_image.getHeight();

        if (cmd == "status") {
            std::printf("{\"status\": \"%s\"}\n", codeName(code));
            return 0;
        }
        if (code != kpeg::JPEGDecoder::DECODE_DONE) {
            std::printf("{\"status\": \"%s\"}\n", codeName(code));
            return 1;
        }
        if (cmd == "decode") {
            dec.dumpRawData();
            std::printf("{\"status\": \"DECODE_DONE\", \"width\": %u, \"height\": %u, "
                        "\"decode_s\": %.6f, \"mpix_per_s\": %.6f}\n",
                        w, h, sec, (double)w * h / sec / 1e6);
        } else if (cmd == "stages") {
            if (argc < 4) return 2;
            std::FILE* f = std::fopen(argv[3], "wb");
            if (!f) return 1;
            unsigned n = (unsigned)dec.m_MCU.size();
            std::fwrite(&w, 4, 1, f);
            std::fwrite(&h, 4, 1, f);
            std::fwrite(&n, 4, 1, f);
            for (auto& m : dec.m_MCU)
                for (int c = 0; c < 3; ++c)
                    for (int r = 0; r < 8; ++r)
                        std::fwrite(m.icoeffs[c][r].data(), sizeof(float), 8, f);
            for (auto& m : dec.m_MCU)
                for (int c = 0; c < 3; ++c)
                    for (int r = 0; r < 8; ++r)
                        std::fwrite(m.m_8x8block[c][r].data(), sizeof(int), 8, f);
            std::fclose(f);
            std::printf("{\"status\": \"DECODE_DONE\", \"width\": %u, \"height\": %u, \"mcus\": %u}\n", w, h, n);
        } else {
            return 2;
        }
    } catch (std::exception& e) {
        // main.cpp:133-137 swallows exceptions the same way (e.g. substr out_of_range
        // on truncated streams, Decoder.cpp:718).
        std::printf("{\"status\": \"EXCEPTION\", \"what\": \"%s\"}\n", e.what());
        rc = 1;
    }
    return rc;
}
