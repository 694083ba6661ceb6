// include/kpeg/Batch.hpp -- many files in one go (extension: SURVEY.md 8(f) item 3; the reference's front end,
// src/main.cpp:19-33, decodes one file per process).
//
// Every file goes through kpeg::JPEGDecoder's marker parser exactly as `kpeg <file.jpg>` would; files whose
// geometry and tables agree are then handed to the GPU path together (kpeg_hip_decode_batch: one upload and one
// fused set of kernel launches per chunk of images instead of six launches per image), the others one by one.
// Every decoded file gets its PPM next to it, named as JPEGDecoder::dumpRawData names it.
#ifndef KPEG_BATCH_HPP
#define KPEG_BATCH_HPP

#include <string>
#include <vector>

namespace kpeg
{
    struct BatchResult
    {
        std::size_t written = 0;    ///< PPM files written
        std::size_t rejected = 0;   ///< names or streams the single-file front end would not have decoded either
        std::size_t failed = 0;     ///< accepted by the parser, but the entropy-coded data is corrupt
        std::size_t groups = 0;     ///< calls into the GPU path (groups of identical geometry and tables)
    };

    /// `names`: files, or directories (their *.jpg entries are taken in name order).  The flags are JPEGDecoder's extensions
    /// (setRestartMarkerSupport, setGrayscaleSupport, setAnySizeSupport, set420Support), all off by default.
    BatchResult decodeFiles( const std::vector<std::string>& names, bool allowDRI = false, bool allowGray = false, bool allowAnySize = false,
                             bool allow420 = false );
}

#endif
