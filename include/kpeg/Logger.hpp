// include/kpeg/Logger.hpp -- the reference's logging surface (include/Logger.hpp:39-126):
// LOG(level) macro, singleton Logger, TeeBuf/TeeStream.  Unlike the reference the level and
// the stream have defined defaults (ERROR, std::clog), so LOG is safe before configuration.
// The GPU path emits no per-MCU lines; parity is defined on the PPM, not on the log.
#ifndef KPEG_LOGGER_HPP
#define KPEG_LOGGER_HPP

#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>

#define __FILENAME__ (strrchr(__FILE__, '/') ? strrchr(__FILE__, '/') + 1 : __FILE__)

#define LOG(level) \
    if (level > kpeg::Logger::get().getLevel()) ;        \
else kpeg::Logger::get().getStream() << kpeg::Logger::levelStr(level) \
                                     << "[ kpeg:"         \
                                     << __FILENAME__     \
                                     << ":" << std::dec  \
                                     << __LINE__ << " ] "

namespace kpeg
{
    class Logger
    {
        public:
            enum Level { ERROR, INFO, DEBUG };

            static inline const std::string levelStr( Level lvl )
            {
                switch ( lvl )
                {
                    case ERROR: return "[ ERROR ]";
                    case INFO:  return "[ INFO  ]";
                    case DEBUG: return "[ DEBUG ]";
                }
                return "";
            }

            ~Logger();
            void setLogStream( std::ostream& stream );
            Logger& setLevel( Level level );
            Level getLevel();
            std::ostream& getStream();
            static Logger& get();

        private:
            Logger();
            Level logLevel_;
            std::ostream* logStream_;
            static std::unique_ptr<Logger> instance_;
    };

    /// streambuf that forwards every character to two other streambufs
    class TeeBuf : public std::streambuf
    {
        public:
            TeeBuf( std::streambuf* sb1, std::streambuf* sb2 );
        private:
            virtual int overflow( int c );
            virtual int sync();
            std::streambuf* sb1_;
            std::streambuf* sb2_;
    };

    class TeeStream : public std::ostream
    {
        public:
            TeeStream( std::ostream& o1, std::ostream& o2 );
        private:
            TeeBuf tbuf_;
    };
}

#endif
