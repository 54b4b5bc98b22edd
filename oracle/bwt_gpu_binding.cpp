// bwt_gpu_binding.cpp -- TEST INFRASTRUCTURE: the binding INTEGRATION.md section 3b shows a maintainer -- libbsc's bsc_bwt_encode
// (libbsc/bwt/bwt.cpp:43-74, the call bsc_compress makes per block, libbsc/libbsc/libbsc.cpp:286) served by nsgpu_bwt_block over
// libnsgpu.so -- compiled into oracle/_ref/backendref_gpu beside the reference's own libbsc (oracle/Makefile: bwt.cpp is compiled where it
// lies with its own encoder renamed, so that bsc_bwt_decode and everything else stay the reference's).  tests/test_bwt_gpu.py checks that
// the .bsc files this binary writes are the reference's byte for byte and decode with the reference's decoder.
#include <cstdint>
#include <cstdio>
#include <vector>
#include "nsgpu.h"
#include "libbsc.h"
#include "bwt/bwt.h"                         // (the declaration decides the linkage of what is defined below)

nsgpu_ctx *g_nsgpu = nullptr;                    // created by the driver (ref_backend_driver.cpp, -DNSGPU_BWT)
double g_bwt_gpu_ms = 0;                         // device time of all blocks (reported by the driver)
unsigned long long g_bwt_blocks = 0, g_bwt_bytes = 0;

int bsc_bwt_encode(unsigned char *T, int n, unsigned char *num_indexes, int *indexes, int /*features*/)
{
    int mod = n / 8;                             // the sampling rate of the auxiliary indexes, as bwt.cpp:50-56 derives it
    mod |= mod >> 1; mod |= mod >> 2; mod |= mod >> 4; mod |= mod >> 8; mod |= mod >> 16; mod >>= 1;
    std::vector<int32_t> aux((size_t)(n - 1) / (mod + 1) + 2);
    int32_t primary = 0;
    uint32_t n_aux = 0;
    double ms = 0;
    if (nsgpu_bwt_block(g_nsgpu, T, (uint64_t)n, T, &primary, (uint32_t)(mod + 1), aux.data(), &n_aux, &ms, nullptr) != NSGPU_OK) {
        fprintf(stderr, "bwt_gpu_binding: %s\n", nsgpu_last_error());
        return LIBBSC_NOT_ENOUGH_MEMORY;         // (the library never falls back to the CPU)
    }
    g_bwt_gpu_ms += ms, ++g_bwt_blocks, g_bwt_bytes += (unsigned long long)n;
    if (num_indexes && indexes) {
        num_indexes[0] = (unsigned char)((n - 1) / (mod + 1));
        for (int t = 0; t < num_indexes[0]; ++t) indexes[t] = aux[t + 1] - 1;
    }
    return primary;
}
