// Logger singleton and tee stream (surface of the reference's src/Logger.cpp).
#include "Logger.hpp"

namespace kpeg
{
    std::unique_ptr<Logger> Logger::instance_ = nullptr;

    Logger::Logger() : logLevel_( Level::ERROR ), logStream_( &std::clog ) {}
    Logger::~Logger() {}

    Logger& Logger::get()
    {
        if ( !instance_ )
            instance_.reset( new Logger );
        return *instance_;
    }

    std::ostream& Logger::getStream() { return *logStream_; }
    void Logger::setLogStream( std::ostream& stream ) { logStream_ = &stream; }

    Logger& Logger::setLevel( Level level )
    {
        logLevel_ = level;
        return *this;
    }

    Logger::Level Logger::getLevel() { return logLevel_; }

    TeeBuf::TeeBuf( std::streambuf* sb1, std::streambuf* sb2 ) : sb1_( sb1 ), sb2_( sb2 ) {}

    int TeeBuf::overflow( int c )
    {
        if ( c == EOF )
            return !EOF;
        const int a = sb1_->sputc( (char)c );
        const int b = sb2_->sputc( (char)c );
        return ( a == EOF || b == EOF ) ? EOF : c;
    }

    int TeeBuf::sync()
    {
        const int a = sb1_->pubsync();
        const int b = sb2_->pubsync();
        return ( a == 0 && b == 0 ) ? 0 : -1;
    }

    TeeStream::TeeStream( std::ostream& o1, std::ostream& o2 ) : std::ostream( &tbuf_ ), tbuf_( o1.rdbuf(), o2.rdbuf() ) {}
}
