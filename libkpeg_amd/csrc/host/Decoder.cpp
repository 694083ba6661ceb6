// kpeg::JPEGDecoder -- host-side JFIF marker parser + the call into the GPU path.
//
// The parser reproduces the accept/reject behaviour of the reference's src/Decoder.cpp
// (catalogued in SURVEY.md A.1), including the parts that look like bugs, because they
// decide which files produce a PPM at all:
//   * every marker must be introduced by exactly one FF; an unknown marker is "accepted"
//     WITHOUT skipping its payload, so the next payload byte != FF ends the decode with
//     ERROR (that is how DRI, APP1..15, SOF3.. are rejected)            (:53-75, :105-133)
//   * APP0 ignores its length field and skips 3*Xthumb*Ythumb bytes                (:164-228)
//   * DQT: (len-2)/65 tables, stored by blind push_back, so ids must arrive 0, 1   (:230-299)
//   * SOF0 reads exactly three component triples; sampling != 1x1 -> TERMINATE     (:301-364)
//   * DHT: several tables per segment, ids 0/1 only                                (:366-459)
//   * SOS: selectors ignored, 3 bytes skipped, then the scan runs to FF D9         (:461-577)
// The file is read into memory once; the byte cursor keeps std::ifstream's "a failed
// extraction leaves the variable unchanged" semantics that the reference relies on.
#include "Decoder.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>

#include "HipContext.hpp"
#include "Logger.hpp"
#include "Markers.hpp"
#include "Transform.hpp"
#include "Utility.hpp"

namespace kpeg
{
    JPEGDecoder::JPEGDecoder() :
        pos_( 0 ), eof_( false ), isOpen_( false ), tableBroken_( false ), sosCount_( 0 ), restartInterval_( 0 ),
        allowDRI_( false ), parseOnly_( false )
    {
        LOG(Logger::Level::INFO) << "Created \'JPEGDecoder object\'." << std::endl;
    }

    JPEGDecoder::JPEGDecoder( const std::string& ) : JPEGDecoder() {}

    JPEGDecoder::~JPEGDecoder()
    {
        hip::releaseResident( &image_ );   // (pixels left on the GPU die with the decoder)
        close();
        LOG(Logger::Level::INFO) << "Destroyed \'JPEGDecoder object\'." << std::endl;
    }

    bool JPEGDecoder::open( const std::string& filename )
    {
        std::ifstream in( filename, std::ios::in | std::ios::binary );
        if ( !in.is_open() || !in.good() )
        {
            LOG(Logger::Level::ERROR) << "Unable to open image: \'" + filename + "\'" << std::endl;
            return false;
        }
        in.seekg( 0, std::ios::end );
        const std::streamoff n = in.tellg();
        in.seekg( 0, std::ios::beg );
        file_.resize( n > 0 ? (std::size_t)n : 0 );
        if ( n > 0 )
            in.read( reinterpret_cast<char*>( file_.data() ), n );
        pos_ = 0;
        eof_ = false;
        isOpen_ = true;
        filename_ = filename;
        LOG(Logger::Level::INFO) << "Opened JPEG image: \'" + filename + "\'" << std::endl;
        return true;
    }

    void JPEGDecoder::openMemory( const UInt8* data, std::size_t size, const std::string& name )
    {
        file_.assign( data, data + size );
        pos_ = 0;
        eof_ = false;
        isOpen_ = true;
        filename_ = name;
    }

    void JPEGDecoder::close()
    {
        isOpen_ = false;
        LOG(Logger::Level::INFO) << "Closed image file: \'" + filename_ + "\'" << std::endl;
    }

    // ---- byte cursor ---------------------------------------------------------------------
    bool JPEGDecoder::readByte( UInt8& b )
    {
        if ( eof_ || pos_ >= file_.size() )
        {
            eof_ = true;
            return false;
        }
        b = file_[pos_++];
        return true;
    }

    UInt16 JPEGDecoder::readBE16()
    {
        UInt8 hi = 0, lo = 0;
        readByte( hi );
        readByte( lo );
        return (UInt16)( ( hi << 8 ) | lo );
    }

    void JPEGDecoder::skip( std::size_t n )
    {
        if ( eof_ )
            return;
        pos_ = ( pos_ + n > file_.size() ) ? file_.size() : pos_ + n;
    }

    // ---- marker dispatch -------------------------------------------------------------------
    JPEGDecoder::ResultCode JPEGDecoder::parseSegmentInfo( const UInt8 byte )
    {
        if ( byte == JFIF_BYTE_0 || byte == JFIF_BYTE_FF )
            return ERROR;

        switch ( byte )
        {
            case JFIF_SOI:  LOG(Logger::Level::INFO) << "Found segment, Start of Image (FFD8)" << std::endl; return SUCCESS;
            case JFIF_APP0: LOG(Logger::Level::INFO) << "Found segment, JPEG/JFIF Image Marker segment (APP0)" << std::endl; parseJFIFSegment(); return SUCCESS;
            case JFIF_COM:  LOG(Logger::Level::INFO) << "Found segment, Comment(FFFE)" << std::endl; parseComment(); return SUCCESS;
            case JFIF_DQT:  LOG(Logger::Level::INFO) << "Found segment, Define Quantization Table (FFDB)" << std::endl; parseQuantizationTable(); return SUCCESS;
            case JFIF_SOF0: LOG(Logger::Level::INFO) << "Found segment, Start of Frame 0: Baseline DCT (FFC0)" << std::endl; return parseSOF0Segment();
            case JFIF_SOF1: LOG(Logger::Level::INFO) << "Found segment, Start of Frame 1: Extended Sequential DCT (FFC1), Not supported" << std::endl; return TERMINATE;
            case JFIF_SOF2: LOG(Logger::Level::INFO) << "Found segment, Start of Frame 2: Progressive DCT (FFC2), Not supported" << std::endl; return TERMINATE;
            case JFIF_DHT:  LOG(Logger::Level::INFO) << "Found segment, Define Huffman Table (FFC4)" << std::endl; parseHuffmanTable(); return SUCCESS;
            case JFIF_SOS:  LOG(Logger::Level::INFO) << "Found segment, Start of Scan (FFDA)" << std::endl; parseSOSSegment(); return SUCCESS;
            case JFIF_DRI:
                if ( allowDRI_ )
                {
                    LOG(Logger::Level::INFO) << "Found segment, Define Restart Interval (FFDD) [extension]" << std::endl;
                    parseDRISegment();
                }
                return SUCCESS;  // reference: unknown marker, payload not skipped
        }
        return SUCCESS;
    }

    JPEGDecoder::ResultCode JPEGDecoder::decodeImageFile()
    {
        if ( !isOpen_ )
        {
            LOG(Logger::Level::ERROR) << "Unable scan image file: \'" + filename_ + "\'" << std::endl;
            return ERROR;
        }
        LOG(Logger::Level::INFO) << "Started decoding process..." << std::endl;

        UInt8 byte = 0;
        ResultCode status = DECODE_DONE;
        while ( readByte( byte ) )
        {
            if ( byte != JFIF_BYTE_FF )
            {
                LOG(Logger::Level::ERROR) << "[ FATAL ] Invalid JFIF file! Terminating..." << std::endl;
                status = ERROR;
                break;
            }
            readByte( byte );
            const ResultCode code = parseSegmentInfo( byte );
            if ( code == TERMINATE ) { status = TERMINATE; break; }
            if ( code == DECODE_INCOMPLETE ) { status = DECODE_INCOMPLETE; break; }
            // SUCCESS continues; ERROR (FF00 / FFFF where a marker is expected) is not acted upon
            // by the reference's if/else chain either (src/Decoder.cpp:113-124)
        }

        if ( status == DECODE_DONE )
        {
            if ( parseOnly_ )
                return status;
            const ResultCode rc = decodeScanData();
            if ( rc != SUCCESS )
                return rc;
            LOG(Logger::Level::INFO) << "Finished decoding process [OK]." << std::endl;
        }
        else if ( status == TERMINATE )
        {
            LOG(Logger::Level::INFO) << "Terminated decoding process [NOT-OK]." << std::endl;
        }
        else if ( status == DECODE_INCOMPLETE )
        {
            LOG(Logger::Level::INFO) << "Decoding process incomplete [NOT-OK]." << std::endl;
        }
        return status;
    }

    bool JPEGDecoder::dumpRawData()
    {
        std::size_t extPos = filename_.find( ".jpg" );
        if ( extPos == std::string::npos )
            extPos = filename_.find( ".jpeg" );
        const std::string target = filename_.substr( 0, extPos ) + ".ppm";
        image_.dumpRawData( target );
        return true;
    }

    // ---- segments ----------------------------------------------------------------------------
    void JPEGDecoder::parseJFIFSegment()
    {
        UInt8 b = 0, xThumb = 0, yThumb = 0, major = 0, minor = 0;
        (void)readBE16();  // length: read, never used to skip
        skip( 5 );         // "JFIF\0"
        readByte( major );
        readByte( minor );
        image_.setJPEGVersion( std::to_string( major ) + "." + std::to_string( minor >> 4 ) + std::to_string( minor & 0x0F ) );
        readByte( b );     // density unit
        (void)readBE16();  // x density
        (void)readBE16();  // y density
        readByte( xThumb );
        readByte( yThumb );
        skip( (std::size_t)3 * xThumb * yThumb );
    }

    void JPEGDecoder::parseComment()
    {
        const UInt16 len = readBE16();
        std::string comment;
        UInt8 b = 0;
        for ( int i = 0; i < (int)len - 2; ++i )
        {
            readByte( b );
            if ( b == JFIF_BYTE_FF )
            {
                LOG(Logger::Level::ERROR) << "Unexpected start of marker at offest: " << pos_ << std::endl;
                return;  // gives up mid-segment without storing the comment
            }
            comment.push_back( (char)b );
        }
        image_.setComment( comment );
    }

    void JPEGDecoder::parseQuantizationTable()
    {
        UInt16 len = readBE16();
        len = (UInt16)( len - 2 );
        for ( int t = 0; t < (int)len / 65; ++t )
        {
            UInt8 pqtq = 0, q = 0;
            readByte( pqtq );
            const std::size_t id = pqtq & 0x0F;
            QTables_.push_back( {} );
            if ( id >= QTables_.size() )
            {
                // the reference indexes past the vector here (undefined behaviour)
                LOG(Logger::Level::ERROR) << "Quantization table id " << id << " before table " << QTables_.size() - 1
                                          << ": outside the supported layout" << std::endl;
                tableBroken_ = true;
                for ( int i = 0; i < 64; ++i ) readByte( q );
                continue;
            }
            for ( int i = 0; i < 64; ++i )
            {
                readByte( q );
                QTables_[id].push_back( (UInt16)q );
            }
        }
    }

    JPEGDecoder::ResultCode JPEGDecoder::parseSOF0Segment()
    {
        UInt8 b = 0, id = 0, samp = 0, tq = 0;
        (void)readBE16();
        readByte( b );  // precision
        const UInt16 h = readBE16();
        const UInt16 w = readBE16();
        readByte( b );  // component count (the reference's loop always reads three triples)
        // extension: a one-component frame is read as what it is (and decoded as grayscale)
        components_ = ( allowGray_ && b == 1 ) ? 1 : 3;
        bool nonSampled = true, is420 = components_ == 3;
        sub420_ = false;
        for ( int i = 0; i < components_; ++i )
        {
            readByte( id );
            readByte( samp );
            readByte( tq );
            if ( ( samp >> 4 ) != 1 || ( samp & 0x0F ) != 1 )
                nonSampled = false;
            // extension: luma 2x2 with quantiser 0, both chroma components 1x1 with quantiser 1 (the reference's hard-wired tables)
            if ( samp != ( i == 0 ? 0x22 : 0x11 ) || tq != ( i == 0 ? 0 : 1 ) )
                is420 = false;
        }
        if ( !nonSampled && allow420_ && is420 )
        {
            sub420_ = true;
            nonSampled = true;
        }
        if ( !nonSampled )
        {
            LOG(Logger::Level::INFO) << "Chroma subsampling not yet supported!" << std::endl;
            return TERMINATE;
        }
        image_.setDimensions( w, h );
        return SUCCESS;
    }

    void JPEGDecoder::parseHuffmanTable()
    {
        const UInt16 len = readBE16();
        const std::size_t segmentEnd = pos_ + len - 2;
        while ( !eof_ && pos_ < segmentEnd )
        {
            UInt8 info = 0, c = 0;
            readByte( info );
            const int cls = ( info & 0x10 ) >> 4, id = info & 0x0F;
            if ( id > 1 )
            {
                LOG(Logger::Level::ERROR) << "Huffman table id " << id << " is outside the supported layout" << std::endl;
                tableBroken_ = true;
                return;
            }
            HuffmanTable& t = huffmanTable_[cls][id];
            int total = 0;
            for ( int i = 0; i < 16; ++i )
            {
                readByte( c );
                t[i].first = c;
                total += c;
            }
            // symbols are handed to the code lengths in order
            int li = 0;
            for ( int s = 0; s < total; ++s )
            {
                readByte( c );
                while ( li < 16 && (int)t[li].second.size() >= t[li].first )
                    ++li;
                if ( li == 16 )
                {
                    tableBroken_ = true;  // redefinition of a table: the reference keeps appending
                    break;
                }
                t[li].second.push_back( c );
            }
            huffmanTree_[cls][id].constructHuffmanTree( t );
        }
    }

    void JPEGDecoder::parseDRISegment()
    {
        (void)readBE16();
        restartInterval_ = readBE16();
    }

    void JPEGDecoder::parseSOSSegment()
    {
        UInt8 n = 0, b = 0;
        (void)readBE16();
        readByte( n );
        if ( n < 1 || n > 4 )
        {
            LOG(Logger::Level::ERROR) << "Invalid component count in image scan: " << (int)n << ", terminating decoding process..." << std::endl;
            return;
        }
        for ( int i = 0; i < n; ++i )
            (void)readBE16();  // component id + table selectors: ignored (tables are hard-wired)
        for ( int i = 0; i < 3; ++i )
            readByte( b );
        sosCount_++;
        scanImageData();
    }

    // everything up to FF D9; an FF followed by anything else keeps both bytes
    void JPEGDecoder::scanImageData()
    {
        UInt8 b = 0;
        scan_.reserve( scan_.size() + ( file_.size() - pos_ ) );
        while ( readByte( b ) )
        {
            if ( b == JFIF_BYTE_FF )
            {
                const UInt8 prev = b;
                readByte( b );  // at end of file b stays FF
                if ( b == JFIF_EOI )
                {
                    LOG(Logger::Level::INFO) << "Found segment, End of Image (FFD9)" << std::endl;
                    return;
                }
                scan_.push_back( prev );
            }
            scan_.push_back( b );
        }
    }

    // ---- the seam ------------------------------------------------------------------------------
    bool JPEGDecoder::frameInfo( kpeg_frame* f ) const
    {
        const bool gray = components_ == 1;   // one quantiser and one pair of Huffman tables: id 0 stands in for id 1
        if ( !f || tableBroken_ || QTables_.size() < ( gray ? 1u : 2u ) || QTables_[0].size() < 64 || ( !gray && QTables_[1].size() < 64 ) )
            return false;
        std::memset( f, 0, sizeof( *f ) );
        f->width = image_.getWidth();
        f->height = image_.getHeight();
        f->components = gray ? 1 : ( sub420_ ? KPEG_FRAME_420 : 0 );
        for ( int t = 0; t < 2; ++t )
            for ( int k = 0; k < 64; ++k )
                f->qt[t][k] = QTables_[gray ? 0 : t][k];  // first 64 entries: what MCU.cpp:110-112 reads
        for ( int cls = 0; cls < 2; ++cls )
            for ( int id = 0; id < 2; ++id )
            {
                int k = 0;
                for ( int i = 0; i < 16; ++i )
                {
                    const auto& e = huffmanTable_[cls][gray ? 0 : id][i];
                    if ( e.first < 0 || e.first > 255 || (int)e.second.size() != e.first || k + e.first > 256 )
                        return false;
                    f->dht[cls][id].counts[i] = (UInt8)e.first;
                    for ( UInt8 s : e.second )
                        f->dht[cls][id].symbols[k++] = s;
                }
                if ( k == 0 )
                    return false;
            }
        f->restart_interval = restartInterval_;
        return true;
    }

    bool JPEGDecoder::decodable() const
    {
        kpeg_frame f;
        const unsigned w = image_.getWidth(), h = image_.getHeight();
        return !scan_.empty() && sosCount_ == 1 && frameInfo( &f ) && w != 0 && h != 0 && ( allowAnySize_ || sub420_ || ( !( w & 7 ) && !( h & 7 ) ) );
    }

    JPEGDecoder::ResultCode JPEGDecoder::decodeScanData()
    {
        if ( scan_.empty() )
        {
            LOG(Logger::Level::ERROR) << " [ FATAL ] Invalid image scan data" << std::endl;
            return SUCCESS;  // the reference logs and still reports DECODE_DONE with an empty image
        }
        kpeg_frame f;
        const unsigned w = image_.getWidth(), h = image_.getHeight();
        if ( sosCount_ != 1 || !frameInfo( &f ) || w == 0 || h == 0 || ( !allowAnySize_ && !sub420_ && ( ( w & 7 ) || ( h & 7 ) ) ) )
        {
            LOG(Logger::Level::ERROR) << "[ FATAL ] Stream is outside what libKPEG decodes without undefined behaviour "
                                         "(two quantisation tables id 0,1; four Huffman tables id 0/1; one scan; "
                                         "dimensions multiples of 8)" << std::endl;
            return ERROR;
        }
        std::string why;
        kpeg_hip_ctx* ctx = hip::context( &why );
        if ( !ctx )
        {
            LOG(Logger::Level::ERROR) << "[ FATAL ] " << why << std::endl;
            return ERROR;
        }
        int rc;
        const std::vector<kpeg_hip_ctx*>& many = f.restart_interval ? hip::contexts( &why ) : std::vector<kpeg_hip_ctx*>();
        if ( many.size() > 1 && !sub420_ && !( w & 7 ) && !( h & 7 ) && ( w / 8 ) % f.restart_interval == 0 )   // every MCU row starts a restart interval
        {
            // extension (the reference rejects DRI): a restart-interval image goes over $KPEG_HIP_DEVICES GPUs as row stripes,
            // every GPU downloads its own rows
            std::vector<UInt8> rgb( (std::size_t)w * h * 3 );
            rc = kpeg_hip_decode_sharded( many.data(), (int)many.size(), &f, scan_.data(), scan_.size(), rgb.data() );
            if ( rc == KPEG_HIP_OK )
                image_.adoptRGB8( std::move( rgb ) );
        }
        else
        {
            // The pixels stay on the GPU until somebody asks for them (Image::setLazySource): dumpRawData() then streams
            // them to the file in bands through pinned buffers.  A later decode on the shared context would overwrite
            // them: whoever decodes next fetches a still pending image first (hip::claimResident).
            hip::claimResident( &image_ );
            rc = kpeg_hip_decode_scan_resident( ctx, &f, scan_.data(), scan_.size() );
            if ( rc == KPEG_HIP_OK )
            {
                Image* self = &image_;
                // (the generation of the pixels this decode left in the context's buffer: a later decode bumps it, and a source that
                // somebody kept beyond that -- dumpRawData() leaves it in place -- answers "gone" instead of another picture's pixels)
                const unsigned long long gen = kpeg_hip_resident_generation( ctx );
                image_.setLazySource( [ctx, f, self, gen]( const Image::BandSink& sink ) {
                    struct Tramp
                    {
                        static int call( void* user, uint32_t row0, uint32_t rows, const uint8_t* p, size_t )
                        {
                            return ( *static_cast<const Image::BandSink*>( user ) )( row0, rows, p ) ? 0 : 1;
                        }
                    };
                    (void)self;
                    if ( kpeg_hip_resident_generation( ctx ) != gen )
                        return false;
                    return kpeg_hip_download_bands( ctx, &f, 0, &Tramp::call, const_cast<Image::BandSink*>( &sink ) ) == KPEG_HIP_OK;
                } );
            }
            else
                hip::releaseResident( &image_ );
        }
        if ( rc != KPEG_HIP_OK )
        {
            LOG(Logger::Level::ERROR) << "[ FATAL ] GPU decode failed: " << kpeg_hip_strerror( rc ) << ": " << kpeg_hip_last_error( ctx ) << std::endl;
            return ERROR;
        }
        return SUCCESS;
    }
}
