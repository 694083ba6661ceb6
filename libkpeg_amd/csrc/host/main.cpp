// kpeg -- command line front end: `kpeg <file.jpg>` writes <file>.ppm.
//
// Same contract as the reference's main.cpp:19-79 for the decode direction: the file name
// must end in ".jpg" (isValidFilename), the PPM lands next to the input, kpeg.log is created
// in the working directory.  The encode direction (`kpeg in.ppm out.jpg`) belongs to the
// reference's unfinished encoder and is out of scope here; it reports that and exits.
// Extensions: `--allow-dri` accepts streams with restart markers, `--allow-gray` one-component files, `--allow-any-size` widths and heights that are not multiples of 8, `--allow-420` 4:2:0 files; `--batch` takes any number of files and
// directories and decodes files of identical geometry and tables together (kpeg::decodeFiles).
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include <vector>

#include "Batch.hpp"
#include "Decoder.hpp"
#include "Logger.hpp"
#include "Utility.hpp"

static void printHelp()
{
    std::cout << "===========================================" << std::endl;
    std::cout << "   K-PEG - Simple JPEG Decoder (MI355X)"      << std::endl;
    std::cout << "===========================================" << std::endl;
    std::cout << "Help\n" << std::endl;
    std::cout << "<filename.jpg>                  : Decompress a JPEG image to a PPM image" << std::endl;
    std::cout << "--allow-dri <filename.jpg>      : same, accepting restart markers (extension)" << std::endl;
    std::cout << "--batch [--allow-dri] [--allow-gray] [--allow-any-size] [--allow-420] <files and directories...> : Decompress many JPEG images (extension)" << std::endl;
    std::cout << "--allow-gray <filename.jpg>     : ... accepting one-component (grayscale) files (extension)" << std::endl;
    std::cout << "--allow-any-size <filename.jpg> : ... accepting widths and heights that are not multiples of 8 (extension)" << std::endl;
    std::cout << "--allow-420 <filename.jpg>      : ... accepting 4:2:0 files of any size (extension)" << std::endl;
    std::cout << "-h                              : Print this help message and exit" << std::endl;
}

static int decodeJPEG( const std::string& filename, bool allowDRI, bool allowGray = false, bool allowAnySize = false, bool allow420 = false )
{
    if ( !kpeg::isValidFilename( filename ) )
    {
        LOG(kpeg::Logger::Level::ERROR) << "Invalid input file name passed." << std::endl;
        return EXIT_SUCCESS;  // the reference returns success here too (main.cpp:21-25,73)
    }
    kpeg::JPEGDecoder decoder;
    decoder.setRestartMarkerSupport( allowDRI );
    decoder.setGrayscaleSupport( allowGray );
    decoder.setAnySizeSupport( allowAnySize );
    decoder.set420Support( allow420 );
    decoder.open( filename );
    if ( decoder.decodeImageFile() == kpeg::JPEGDecoder::ResultCode::DECODE_DONE )
        decoder.dumpRawData();
    return EXIT_SUCCESS;
}

#ifndef KPEG_SRC_HASH
#define KPEG_SRC_HASH "unstamped"
#endif
// libkpeg_amd/build.py looks for this string in the binary and rebuilds when the sources have changed
extern "C" const char kpeg_cli_build_stamp[] = "KPEG_SRC_HASH=" KPEG_SRC_HASH;

int main( int argc, char** argv )
{
    try
    {
        std::ofstream logFile( "kpeg.log", std::ios::out );
        kpeg::TeeStream logTee( logFile, std::cout );
        if ( logFile.is_open() && logFile.good() )
            kpeg::Logger::get().setLogStream( logTee );
        else
            kpeg::Logger::get().setLogStream( std::cout );
        kpeg::Logger::get().setLevel( kpeg::Logger::Level::DEBUG );
        LOG(kpeg::Logger::Level::INFO) << "KPEG - Simple JPEG Decoder (MI355X path)" << std::endl;

        if ( argc < 2 )
        {
            LOG(kpeg::Logger::Level::ERROR) << "No arguments provided." << std::endl;
            return EXIT_FAILURE;
        }
        if ( argc == 2 && std::string( argv[1] ) == "-h" )
        {
            printHelp();
            return EXIT_SUCCESS;
        }
        if ( argc >= 3 && std::string( argv[1] ) == "--batch" )
        {
            bool allowDRI = false, allowGray = false, allowAnySize = false, allow420 = false;
            std::vector<std::string> names;
            for ( int i = 2; i < argc; ++i )
            {
                const std::string arg( argv[i] );
                if ( arg == "--allow-dri" )
                    allowDRI = true;
                else if ( arg == "--allow-gray" )
                    allowGray = true;
                else if ( arg == "--allow-any-size" )
                    allowAnySize = true;
                else if ( arg == "--allow-420" )
                    allow420 = true;
                else
                    names.push_back( arg );
            }
            kpeg::Logger::get().setLevel( kpeg::Logger::Level::ERROR );   // one INFO line per marker and file is too much here
            const kpeg::BatchResult r = kpeg::decodeFiles( names, allowDRI, allowGray, allowAnySize, allow420 );
            std::cout << "kpeg --batch: " << r.written << " PPM written, " << r.rejected << " rejected, " << r.failed
                      << " failed, " << r.groups << " group(s)" << std::endl;
            return r.failed ? EXIT_FAILURE : EXIT_SUCCESS;
        }
        if ( argc == 2 )
            return decodeJPEG( argv[1], false );
        if ( argc == 3 && std::string( argv[1] ) == "--allow-dri" )
            return decodeJPEG( argv[2], true );
        if ( argc == 3 && std::string( argv[1] ) == "--allow-gray" )
            return decodeJPEG( argv[2], false, true );
        if ( argc == 3 && std::string( argv[1] ) == "--allow-any-size" )
            return decodeJPEG( argv[2], false, false, true );
        if ( argc == 3 && std::string( argv[1] ) == "--allow-420" )
            return decodeJPEG( argv[2], false, false, false, true );
        if ( argc == 3 )
        {
            LOG(kpeg::Logger::Level::ERROR) << "The PPM->JPEG encoder of libKPEG is unfinished upstream and is not part of this build." << std::endl;
            return EXIT_SUCCESS;
        }
        return EXIT_FAILURE;
    }
    catch ( std::exception& e )
    {
        std::cout << "Exceptions Occurred:-" << std::endl;
        std::cout << "What: " << e.what() << std::endl;
    }
    return EXIT_SUCCESS;
}
