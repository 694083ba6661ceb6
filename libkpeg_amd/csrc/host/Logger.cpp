// Logger singleton and tee stream (surface of the reference's src/Logger.cpp).
#include "Logger.hpp"

namespace kpeg
{
    std::unique_ptr<Logger> Logger::m_instance = nullptr;

    Logger::Logger() : m_logLevel( Level::ERROR ), m_logStream( &std::clog ) {}
    Logger::~Logger() {}

    Logger& Logger::get()
    {
        if ( !m_instance )
            m_instance.reset( new Logger );
        return *m_instance;
    }

    std::ostream& Logger::getStream() { return *m_logStream; }
    void Logger::setLogStream( std::ostream& stream ) { m_logStream = &stream; }

    Logger& Logger::setLevel( Level level )
    {
        m_logLevel = level;
        return *this;
    }

    Logger::Level Logger::getLevel() { return m_logLevel; }

    TeeBuf::TeeBuf( std::streambuf* sb1, std::streambuf* sb2 ) : m_sb1( sb1 ), m_sb2( sb2 ) {}

    int TeeBuf::overflow( int c )
    {
        if ( c == EOF )
            return !EOF;
        const int a = m_sb1->sputc( (char)c );
        const int b = m_sb2->sputc( (char)c );
        return ( a == EOF || b == EOF ) ? EOF : c;
    }

    int TeeBuf::sync()
    {
        const int a = m_sb1->pubsync();
        const int b = m_sb2->pubsync();
        return ( a == 0 && b == 0 ) ? 0 : -1;
    }

    TeeStream::TeeStream( std::ostream& o1, std::ostream& o2 ) : std::ostream( &m_tbuf ), m_tbuf( o1.rdbuf(), o2.rdbuf() ) {}
}
