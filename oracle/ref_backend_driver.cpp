// ref_backend_driver.cpp -- the reference's back-end coders behind a command line, for MEASUREMENTS of SURVEY 8(f4) only
// (tools/backend_coders.py).  TEST / MEASUREMENT INFRASTRUCTURE: nothing here is restated or shipped -- it calls the reference's own
// bsc::BSC_compress / BSC_decompress (src/bsc.cpp:1045-1069: libbsc, block size 48 MB, -p -e2) and lzma2::lzma2_compress /
// lzma2_decompress (src/lzma2.cpp, fast-lzma2 preset 6), compiled where they lie under /root/reference by oracle/Makefile into
// oracle/_ref/backendref, exactly as Compressor::compress() calls them per stream file (src/Compressor.cpp:111-143).
//   backendref bsc|lzma2|unbsc|unlzma2 <in> <out>
#include <cstdio>
#include <cstring>
#include "bsc_helper.h"
#include "lzma2_helper.h"

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: backendref bsc|lzma2|unbsc|unlzma2 in out\n"); return 2; }
    if (!strcmp(argv[1], "bsc")) bsc::BSC_compress(argv[2], argv[3]);
    else if (!strcmp(argv[1], "lzma2")) lzma2::lzma2_compress(argv[2], argv[3]);
    else if (!strcmp(argv[1], "unbsc")) bsc::BSC_decompress(argv[2], argv[3]);
    else if (!strcmp(argv[1], "unlzma2")) lzma2::lzma2_decompress(argv[2], argv[3]);
    else return 2;
    return 0;
}
