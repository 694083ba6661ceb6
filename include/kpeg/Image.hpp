// include/kpeg/Image.hpp -- decoded image store and PPM I/O.
//
// Same surface as the reference's include/Image.hpp:24-62.  The GPU path delivers packed
// RGB8 rows; the reference's vector<vector<Pixel>> view (Int16 x 3 per pixel, 199 MB at 8K)
// is materialised lazily on the first getPixelPtr() call, and dumpRawData() writes the PPM
// straight from the RGB8 buffer.
#ifndef KPEG_IMAGE_HPP
#define KPEG_IMAGE_HPP

#include <array>
#include <fstream>
#include <iostream>
#include <memory>
#include <functional>
#include <string>
#include <vector>

#include "MCU.hpp"
#include "Types.hpp"

namespace kpeg
{
    class Image
    {
        public:
            Image();
            /// A copy owns its pixels: a source that still has them with the decoder (setLazySource) is asked for them first, so that
            /// no copy depends on device memory which a later decode reuses.
            Image( const Image& other );
            Image& operator=( const Image& other );
            Image( Image&& ) = default;
            Image& operator=( Image&& ) = default;

            /// Tiles MCUs row-major into the pixel store and crops padding (reference
            /// src/Image.cpp:20-86).  Kept for API compatibility; the decoder uses adoptRGB8().
            void createImageFromMCUs( const std::vector<MCU>& MCUVector );

            PixelPtr getPixelPtr();
            FPixelPtr getFlPixelPtr();
            const unsigned getWidth() const;
            const unsigned getHeight() const;

            /// Binary P6 with the reference's fixed comment line (src/Image.cpp:108-140).
            const bool dumpRawData( const std::string& filename );
            const bool readRawData( const std::string& filename );

            void setImageFilename( const std::string& filename );
            void setJPEGVersion( const std::string& version );
            void setComment( const std::string& comment );
            void setDimensions( const std::size_t width, const std::size_t height );

            // ---- additions for the GPU path ----
            /// Takes ownership of height*width*3 bytes, row-major R,G,B.
            void adoptRGB8( std::vector<UInt8>&& rgb );
            const std::vector<UInt8>& getRGB8() const;

            /// A consumer of row bands (first row, rows, rows*width*3 bytes R,G,B); false stops the source.
            typedef std::function<bool( std::size_t, std::size_t, const UInt8* )> BandSink;
            /// The pixels stay where the decoder left them (GPU memory) until somebody wants them: `source` streams them
            /// band by band into a sink.  dumpRawData() hands the bands straight to the file (the next band crosses PCIe
            /// while one is written); getRGB8() / getPixelPtr() assemble them first.  Replaces the reference's
            /// vector<vector<Pixel>> materialisation (src/Image.cpp:49-70), which at GPU speed was most of the time.
            void setLazySource( std::function<bool( const BandSink& )> source );
            /// Fetches the pixels now if they are still with the decoder (idempotent).
            bool materialise() const;

        private:
            std::string  filename_;
            PixelPtr     pixelPtr_;
            FPixelPtr    flPixelPtr_;
            std::string  JPEGversion_;
            std::string  comment_;
            std::size_t  width_;
            std::size_t  height_;
            mutable std::vector<UInt8> rgb8_;
            mutable std::function<bool( const BandSink& )> lazy_;
    };

    const std::string valueToBitString( const Int16 value );
    const Int16 bitStringtoValue( const std::string& bitStr );
    const Int16 getValueCategory( const Int16 value );
}

#endif
