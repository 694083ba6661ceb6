// kpeg::Image -- pixel store, MCU tiling and PPM I/O (surface of the reference's src/Image.cpp).
#include "Image.hpp"

#include <cstdio>
#include <cstring>
#include <memory>

#include "Logger.hpp"

namespace kpeg
{
    Image::Image() :
        filename_{ "" }, pixelPtr_{ nullptr }, flPixelPtr_{ nullptr }, JPEGversion_{ "" }, comment_{ "" },
        width_{ 0 }, height_{ 0 }
    {
    }

    Image::Image( const Image& o ) :
        filename_{ o.filename_ }, pixelPtr_{ nullptr }, flPixelPtr_{ o.flPixelPtr_ }, JPEGversion_{ o.JPEGversion_ }, comment_{ o.comment_ },
        width_{ o.width_ }, height_{ o.height_ }
    {
        o.materialise();      // (the source keeps the pixels it fetched: both own a copy afterwards)
        rgb8_ = o.rgb8_;
        if ( o.pixelPtr_ )
            pixelPtr_ = std::make_shared<std::vector<std::vector<Pixel>>>( *o.pixelPtr_ );
    }

    Image& Image::operator=( const Image& o )
    {
        if ( this != &o )
        {
            Image tmp( o );
            *this = std::move( tmp );
        }
        return *this;
    }

    // MCU n -> tile (n / tilesPerRow, n % tilesPerRow); pixel (8*tr + v, 8*tc + u) = block[.][v][u];
    // columns/rows beyond the image size are cropped (reference src/Image.cpp:26-84).
    void Image::createImageFromMCUs( const std::vector<MCU>& MCUVector )
    {
        const std::size_t tw = ( width_ + 7 ) / 8, th = ( height_ + 7 ) / 8;
        if ( MCUVector.size() < tw * th )
        {
            LOG(Logger::Level::ERROR) << "createImageFromMCUs: " << MCUVector.size() << " MCUs given, " << tw * th << " needed" << std::endl;
            return;
        }
        std::vector<UInt8> rgb( width_ * height_ * 3 );
        for ( std::size_t tr = 0; tr < th; ++tr )
            for ( std::size_t tc = 0; tc < tw; ++tc )
            {
                const CompMatrices& b = MCUVector[tr * tw + tc].getAllMatrices();
                for ( std::size_t v = 0; v < 8 && tr * 8 + v < height_; ++v )
                    for ( std::size_t u = 0; u < 8 && tc * 8 + u < width_; ++u )
                        for ( int c = 0; c < 3; ++c )
                            rgb[( ( tr * 8 + v ) * width_ + tc * 8 + u ) * 3 + c] = (UInt8)b[c][v][u];
            }
        adoptRGB8( std::move( rgb ) );
    }

    void Image::adoptRGB8( std::vector<UInt8>&& rgb )
    {
        rgb8_ = std::move( rgb );
        lazy_ = nullptr;
        pixelPtr_.reset();
    }

    void Image::setLazySource( std::function<bool( const BandSink& )> source )
    {
        rgb8_.clear();
        pixelPtr_.reset();
        lazy_ = std::move( source );
    }

    bool Image::materialise() const
    {
        if ( !lazy_ )
            return true;
        // bands arrive in row order: appended to a reserved vector (no zero fill, one copy)
        rgb8_.clear();
        rgb8_.reserve( width_ * height_ * 3 );
        const std::size_t pitch = width_ * 3;
        auto source = std::move( lazy_ );
        lazy_ = nullptr;
        const bool ok = source( [&]( std::size_t, std::size_t rows, const UInt8* p ) {
            rgb8_.insert( rgb8_.end(), p, p + rows * pitch );
            return true;
        } );
        if ( !ok || rgb8_.size() != width_ * height_ * 3 )
            rgb8_.clear();
        return ok;
    }

    const std::vector<UInt8>& Image::getRGB8() const
    {
        materialise();
        return rgb8_;
    }

    PixelPtr Image::getPixelPtr()
    {
        materialise();
        if ( !pixelPtr_ && rgb8_.size() == width_ * height_ * 3 && !rgb8_.empty() )
        {
            pixelPtr_ = std::make_shared<std::vector<std::vector<Pixel>>>( height_, std::vector<Pixel>( width_ ) );
            const UInt8* p = rgb8_.data();
            for ( auto& row : *pixelPtr_ )
                for ( auto& px : row )
                {
                    px.comp[0] = p[0];
                    px.comp[1] = p[1];
                    px.comp[2] = p[2];
                    p += 3;
                }
        }
        return pixelPtr_;
    }

    FPixelPtr Image::getFlPixelPtr() { return flPixelPtr_; }
    const unsigned Image::getWidth() const { return (unsigned)width_; }
    const unsigned Image::getHeight() const { return (unsigned)height_; }

    const bool Image::dumpRawData( const std::string& filename )
    {
        const bool haveRGB = !rgb8_.empty() && rgb8_.size() == width_ * height_ * 3;
        if ( !haveRGB && pixelPtr_ == nullptr && !lazy_ )
        {
            LOG(Logger::Level::ERROR) << "Unable to create dump file \'" + filename + "\', Invalid pixel pointer" << std::endl;
            return false;
        }
        std::FILE* f = std::fopen( filename.c_str(), "wb" );
        if ( !f )
        {
            LOG(Logger::Level::ERROR) << "Unable to create dump file \'" + filename + "\'." << std::endl;
            return false;
        }
        const bool streaming = lazy_ && !haveRGB;
        if ( streaming )
            std::setvbuf( f, nullptr, _IONBF, 0 );   // whole bands go down in one write each (before any other operation on the stream: ISO C 7.21.5.6)
        // header bytes are part of the bit-exact output (reference src/Image.cpp:124-127)
        std::fprintf( f, "P6\n# PPM dump created using libKPEG: https://github.com/TheIllusionistMirage/libKPEG\n%zu %zu\n255\n",
                      width_, height_ );
        bool ok = true;
        if ( lazy_ && !haveRGB )
        {
            // the pixels are still with the decoder: band k is written while band k + 1 is on its way; nothing is assembled
            const std::size_t pitch = width_ * 3;
            auto source = lazy_;   // (kept: the file is one consumer; a later getPixelPtr() may still want the pixels)
            ok = source( [&]( std::size_t, std::size_t rows, const UInt8* p ) { return std::fwrite( p, 1, rows * pitch, f ) == rows * pitch; } );
        }
        else if ( haveRGB )
            ok = std::fwrite( rgb8_.data(), 1, rgb8_.size(), f ) == rgb8_.size();
        else
            for ( auto&& row : *pixelPtr_ )
                for ( auto&& px : row )
                {
                    const UInt8 b[3] = { (UInt8)px.comp[0], (UInt8)px.comp[1], (UInt8)px.comp[2] };
                    ok = ok && std::fwrite( b, 1, 3, f ) == 3;
                }
        ok = ( std::fclose( f ) == 0 ) && ok;
        if ( ok )
        {
            LOG(Logger::Level::INFO) << "Raw image data dumped to file: \'" + filename + "\'." << std::endl;
        }
        return ok;
    }

    // Binary P6 reader (used by the reference's encoder only; kept for API completeness).
    const bool Image::readRawData( const std::string& filename )
    {
        std::ifstream in( filename, std::ios::in | std::ios::binary );
        if ( !in.is_open() || !in.good() )
        {
            LOG(Logger::Level::ERROR) << "Unable to read PPM file: \'" + filename + "\'" << std::endl;
            return false;
        }
        char magic[2] = { 0, 0 };
        in.read( magic, 2 );
        if ( magic[0] != 'P' || magic[1] != '6' )
        {
            LOG(Logger::Level::ERROR) << "Invalid PPM file: \'" + filename + "\'" << std::endl;
            return false;
        }
        auto skipCommentsAndSpace = [&in]() {
            for ( ;; )
            {
                int c = in.peek();
                if ( c == '#' ) { std::string line; std::getline( in, line ); }
                else if ( c == ' ' || c == '\n' || c == '\r' || c == '\t' ) in.get();
                else break;
            }
        };
        unsigned width = 0, height = 0, maxv = 0;
        skipCommentsAndSpace(); in >> width;
        skipCommentsAndSpace(); in >> height;
        skipCommentsAndSpace(); in >> maxv;
        in.get();  // single whitespace before the raster
        if ( !in.good() || width == 0 || height == 0 )
            return false;
        width_ = width;
        height_ = height;
        flPixelPtr_ = std::make_shared<std::vector<std::vector<FPixel>>>( height, std::vector<FPixel>( width ) );
        std::vector<UInt8> row( (std::size_t)width * 3 );
        for ( unsigned y = 0; y < height; ++y )
        {
            in.read( reinterpret_cast<char*>( row.data() ), row.size() );
            if ( (std::size_t)in.gcount() != row.size() )
                return false;
            for ( unsigned x = 0; x < width; ++x )
                ( *flPixelPtr_ )[y][x] = FPixel( row[x * 3], row[x * 3 + 1], row[x * 3 + 2] );
        }
        filename_ = filename;
        return true;
    }

    void Image::setImageFilename( const std::string& filename ) { filename_ = filename; }
    void Image::setJPEGVersion( const std::string& version ) { JPEGversion_ = version; }
    void Image::setComment( const std::string& comment ) { comment_ = comment; }
    void Image::setDimensions( const std::size_t width, const std::size_t height )
    {
        width_ = width;
        height_ = height;
    }

    // ---- bit-string helpers (reference src/Image.cpp:258-320) ------------------------------
    const Int16 getValueCategory( const Int16 value )
    {
        int a = value < 0 ? -(int)value : (int)value, cat = 0;
        while ( a ) { ++cat; a >>= 1; }
        return (Int16)cat;
    }

    // magnitude bits of a coefficient: the value itself if positive, its one's complement if negative
    const std::string valueToBitString( const Int16 value )
    {
        if ( value == 0 )
            return "";
        const int cat = getValueCategory( value );
        const int bits = value > 0 ? value : ( ( 1 << cat ) - 1 ) + value;
        std::string s( cat, '0' );
        for ( int i = 0; i < cat; ++i )
            if ( bits & ( 1 << ( cat - 1 - i ) ) )
                s[i] = '1';
        return s;
    }

    // JPEG EXTEND: leading '1' -> +binary value, leading '0' -> -(one's complement), "" -> 0
    const Int16 bitStringtoValue( const std::string& bitStr )
    {
        if ( bitStr.empty() )
            return 0;
        int v = 0;
        for ( char ch : bitStr )
            v = ( v << 1 ) | ( ch == '1' ? 1 : 0 );
        if ( bitStr[0] == '1' )
            return (Int16)v;
        return (Int16)( -( ( ( 1 << bitStr.size() ) - 1 ) - v ) );
    }
}
