// include/kpeg/Utility.hpp -- string helpers with the reference's observable behaviour
// (include/Utility.hpp:11-52), including its quirks:
//   * isValidChar() is true for every character (`isprint || != '/' || != '\\'`),
//   * isValidFilename() accepts only names that END in ".jpg"; ".jpeg" names are rejected
//     because the length test adds 4, not 5 (Utility.hpp:23-37).
#ifndef KPEG_UTILITY_HPP
#define KPEG_UTILITY_HPP

#include <cctype>
#include <string>

namespace kpeg
{
    inline const bool isValidChar( const char ch )
    {
        return isprint( ch ) || ch != '/' || ch != '\\';
    }

    inline const bool isValidFilename( const std::string& filename )
    {
        for ( auto&& c : filename )
            if ( !isValidChar( c ) )
                return false;

        std::size_t pos = filename.find( ".jpg" );
        if ( pos != std::string::npos )
            return pos + 4 == filename.size();

        pos = filename.find( ".jpeg" );
        if ( pos == std::string::npos )
            return false;
        return pos + 4 == filename.size();  // never true: ".jpeg" has five characters
    }

    inline const bool isWhiteSpace( const char ch )
    {
        return iscntrl( ch ) || isblank( ch ) || isspace( ch );
    }

    inline const bool isStringWhiteSpace( const std::string& str )
    {
        for ( auto&& c : str )
            if ( !isWhiteSpace( c ) )
                return false;
        return true;
    }
}

#endif
