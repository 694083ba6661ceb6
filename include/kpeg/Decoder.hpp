// include/kpeg/Decoder.hpp -- kpeg::JPEGDecoder, the public decode API.
//
// Same public surface and result codes as the reference's include/Decoder.hpp:26-96:
// open / close / decodeImageFile / parseSegmentInfo / dumpRawData / printCurrPos.  The marker
// parser runs on the host and accepts/rejects exactly what the reference does;
// at the seam where the reference calls decodeScanData() + createImageFromMCUs()
// (src/Decoder.cpp:135-139) this class calls the C ABI of the MI355X path
// (include/kpeg_hip.h: kpeg_hip_decode_scan).  There is no CPU decode path: without a gfx950
// device decodeImageFile() logs an error and returns ResultCode::ERROR.
#ifndef KPEG_DECODER_HPP
#define KPEG_DECODER_HPP

#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "HuffmanTree.hpp"
#include "Image.hpp"
#include "MCU.hpp"
#include "Types.hpp"

struct kpeg_frame;

namespace kpeg
{
    class JPEGDecoder
    {
        public:
            enum ResultCode
            {
                SUCCESS ,
                TERMINATE ,
                ERROR ,
                DECODE_INCOMPLETE ,
                DECODE_DONE
            };

        public:
            ResultCode decodeImageFile();

        public:
            JPEGDecoder();
            /// The argument is ignored, as in the reference (src/Decoder.cpp:18-22).
            JPEGDecoder( const std::string& filename );
            ~JPEGDecoder();

            bool open( const std::string& filename );
            void close();
            ResultCode parseSegmentInfo( const UInt8 byte );
            bool dumpRawData();

            inline void printCurrPos()
            {
                std::cout << "Current file pos: 0x" << std::hex << pos_ << std::endl;
            }

            // ---- additions (off by default, outside the parity contract) ----
            /// Accept DRI / RSTn (the reference rejects them).
            void setRestartMarkerSupport( bool on ) { allowDRI_ = on; }
            /// Extension, off by default: accept one-component (grayscale) baseline files.  The reference reads three
            /// component triples whatever SOF0's count says (src/Decoder.cpp:339) and fails on them.
            void setGrayscaleSupport( bool on ) { allowGray_ = on; }
            /// Extension, off by default: accept widths and heights that are not multiples of 8.  The reference decodes
            /// (w * h) / 64 MCUs and then tiles ceil(w / 8) * ceil(h / 8) of them (src/Decoder.cpp:670, src/Image.cpp:26-68):
            /// it reads past its MCU vector.  Here all MCUs of the padded picture are decoded and the picture is cropped
            /// the way Image::createImageFromMCUs crops it.
            void setAnySizeSupport( bool on ) { allowAnySize_ = on; }
            /// Extension, off by default: accept 4:2:0 files (luma sampled 2x2, any size).  The reference answers TERMINATE on
            /// every sampling factor other than 1x1 (src/Decoder.cpp:339-356).  Decoded through the reference's own per-block
            /// arithmetic, every chroma sample repeated over its 2x2 luma samples.
            void set420Support( bool on ) { allow420_ = on; }
            /// Parse only: stop at the seam and leave the tables for frameInfo().
            void setParseOnly( bool on ) { parseOnly_ = on; }
            /// Tables and geometry as handed to the GPU path; valid after decodeImageFile().
            bool frameInfo( kpeg_frame* out ) const;
            /// The entropy-coded segment as scanImageData collected it (still byte-stuffed).
            const std::vector<UInt8>& scanData() const { return scan_; }
            Image& image() { return image_; }
            /// After a parse-only decodeImageFile(): the stream is one the GPU path takes (two quantisation tables, four
            /// Huffman tables, exactly one non-empty scan, dimensions multiples of 8) -- the conditions decodeScanData() checks.
            bool decodable() const;
            /// Parse an in-memory file instead of open().
            void openMemory( const UInt8* data, std::size_t size, const std::string& name );

        private:
            void parseJFIFSegment();
            void parseQuantizationTable();
            ResultCode parseSOF0Segment();
            void parseHuffmanTable();
            void parseSOSSegment();
            void parseDRISegment();
            void scanImageData();
            void parseComment();
            ResultCode decodeScanData();

            // byte cursor with std::ifstream's "failed reads leave the variable alone" semantics
            bool readByte( UInt8& b );
            UInt16 readBE16();
            void skip( std::size_t n );

        private:
            std::string filename_;
            std::vector<UInt8> file_;
            std::size_t pos_;
            bool eof_;
            bool isOpen_;

            Image image_;
            std::vector<std::vector<UInt16>> QTables_;
            HuffmanTable huffmanTable_[2][2];
            HuffmanTree huffmanTree_[2][2];
            bool tableBroken_;          // a table layout the reference would corrupt memory on

            std::vector<UInt8> scan_;   // entropy-coded segment (reference: scanData_, as bytes)
            int sosCount_;
            UInt32 restartInterval_;
            bool allowDRI_, parseOnly_;
            bool allowGray_ = false;
            bool allowAnySize_ = false;
            bool allow420_ = false;
            bool sub420_ = false;
            int components_ = 3;
    };
}

#endif
