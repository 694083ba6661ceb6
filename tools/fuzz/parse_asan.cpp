#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <fstream>
#include "kpeg_host.h"
int main(int argc, char** argv)
{
    for (int i = 1; i < argc; ++i) {
        std::ifstream f(argv[i], std::ios::binary);
        std::vector<unsigned char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        // the file itself, then every truncation at a few points and a few single-byte corruptions
        for (int variant = 0; variant < 64; ++variant) {
            std::vector<unsigned char> v = d;
            if (variant > 0 && variant < 20 && !v.empty()) v.resize(v.size() * variant / 20);
            if (variant >= 20 && !v.empty()) v[(size_t)(variant - 19) * 7919 % v.size()] ^= (unsigned char)(1u << (variant % 8));
            kpeg_frame fr;
            std::memset(&fr, 0, sizeof(fr));
            std::vector<unsigned char> scan(v.size() + 16);
            size_t n = 0;
            int rc = kpeg_host_parse(v.data(), v.size(), (unsigned)(variant % 16) /* every combination of the four extension flags */, &fr, scan.data(), scan.size(), &n);
            (void)rc;
        }
        std::printf("%s ok\n", argv[i]);
    }
    return 0;
}
