// ref_backend_driver.cpp -- the reference's back-end coders behind a command line, for MEASUREMENTS of SURVEY 8(f4) only
// (tools/backend_coders.py).  TEST / MEASUREMENT INFRASTRUCTURE: nothing here is restated or shipped -- it calls the reference's own
// bsc::BSC_compress / BSC_decompress (src/bsc.cpp:1045-1069: libbsc, block size 48 MB, -p -e2) and lzma2::lzma2_compress /
// lzma2_decompress (src/lzma2.cpp, fast-lzma2 preset 6), compiled where they lie under /root/reference by oracle/Makefile into
// oracle/_ref/backendref, exactly as Compressor::compress() calls them per stream file (src/Compressor.cpp:111-143).
//   backendref bsc|lzma2|unbsc|unlzma2 <in> <out>
//   backendref bwt <in> <out>: libbsc's block sorter alone on the whole file as ONE block -- bsc_bwt_encode (libbsc/bwt/bwt.cpp:46-79), the
//       call bsc_compress makes per block -- dumped as int32 primary index, int32 num_indexes, num_indexes x int32 indexes, n BWT bytes:
//       the pin of nsgpu_bwt_block (tests/test_bwt_gpu.py)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bsc_helper.h"
#include "lzma2_helper.h"
#include "bwt/bwt.h"
#include "libbsc.h"
#ifdef NSGPU_BWT
// backendref_gpu: the same front ends, libbsc's block sorter served by libnsgpu.so (oracle/bwt_gpu_binding.cpp)
#include "nsgpu.h"
extern nsgpu_ctx *g_nsgpu;
extern double g_bwt_gpu_ms;
extern unsigned long long g_bwt_blocks, g_bwt_bytes;
#endif

static int dump_bwt(const char *in, const char *out)
{
    FILE *f = fopen(in, "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> t((size_t)n + 1);
    if (n && fread(t.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); return 1; }
    fclose(f);
    unsigned char num = 0;
    int idx[256];
    memset(idx, 0, sizeof(idx));
    bsc_init(LIBBSC_DEFAULT_FEATURES);
    const int index = n ? bsc_bwt_encode(t.data(), (int)n, &num, idx, LIBBSC_FEATURE_NONE) : 0;
    if (index < 0) return 3;
    FILE *o = fopen(out, "wb");
    if (!o) return 1;
    const int num_i = (int)num;
    fwrite(&index, 4, 1, o);
    fwrite(&num_i, 4, 1, o);
    fwrite(idx, 4, (size_t)num_i, o);
    fwrite(t.data(), 1, (size_t)n, o);
    fclose(o);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: backendref bsc|lzma2|unbsc|unlzma2|bwt in out\n"); return 2; }
#ifdef NSGPU_BWT
    {
        nsgpu_params p;
        nsgpu_default_params(&p);
        if (nsgpu_create(&p, &g_nsgpu) != NSGPU_OK) { fprintf(stderr, "backendref_gpu: %s\n", nsgpu_last_error()); return 4; }
    }
    struct Report { ~Report() { if (g_bwt_blocks) fprintf(stderr, "[backendref_gpu] %llu blocks, %llu bytes through nsgpu_bwt_block, %.1f ms on the device\n", g_bwt_blocks, g_bwt_bytes, g_bwt_gpu_ms); nsgpu_destroy(g_nsgpu); } } report;
#endif
    if (!strcmp(argv[1], "bwt")) return dump_bwt(argv[2], argv[3]);
    if (!strcmp(argv[1], "bsc")) bsc::BSC_compress(argv[2], argv[3]);
    else if (!strcmp(argv[1], "lzma2")) lzma2::lzma2_compress(argv[2], argv[3]);
    else if (!strcmp(argv[1], "unbsc")) bsc::BSC_decompress(argv[2], argv[3]);
    else if (!strcmp(argv[1], "unlzma2")) lzma2::lzma2_decompress(argv[2], argv[3]);
    else return 2;
    return 0;
}
