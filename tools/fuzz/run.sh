#!/bin/bash
# ASan + UBSan build of the host library's marker parser (CPU only), fed with every golden fixture plus
# truncations and single-bit corruptions of each.   tools/fuzz/run.sh
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/asan
g++ -O1 -g -std=c++14 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Iinclude/kpeg -o build/asan/parse_asan \
    tools/fuzz/parse_asan.cpp $(ls libkpeg_amd/csrc/host/*.cpp | grep -v main.cpp) -Llibkpeg_amd -lkpeg_hip -Wl,-rpath,$PWD/libkpeg_amd
ASAN_OPTIONS=detect_leaks=0 ./build/asan/parse_asan tests/golden/*.jpg 2>&1 | grep -v "^\[ ERROR\|^\[ WARN\|^\[ INFO" | grep -v " ok$" || true
echo "sanitizer run finished (lines above, if any, are findings)"
