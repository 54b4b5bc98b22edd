// The adaptor of INTEGRATION.md section 1 as a stand-alone C++11 program: what Compressor::compress() would do with
// libnsgpu.so in place of ReadData / MinHashReadFilter / Consensus (src/Compressor.cpp:57-104), minus the reference's
// own classes (reads come from a plain FASTQ file).  Built and run by tests/test_integration_gpu.py.
//   usage: compress_stage <reads.fastq> <tempDir/> <numThr>
#include "nsgpu.h"
#include <cstdio>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

static void ns_check(int rc) { if (rc != NSGPU_OK) throw std::runtime_error(nsgpu_last_error()); }   // caught in main, as in src/main.cpp:161-176

int main(int argc, char **argv)
{
    if (argc < 4) { std::cerr << "usage: compress_stage <reads.fastq> <tempDir/> <numThr>\n"; return 2; }
    try {
        const std::string tempDir = argv[2], tempFileName = "Stream";
        const uint32_t numThr = (uint32_t)std::stoul(argv[3]);
        std::ifstream in(argv[1], std::ios::binary);
        std::stringstream ss;
        ss << in.rdbuf();
        const std::string text = ss.str();

        nsgpu_params p;
        nsgpu_default_params(&p);                          // k 23, n 60, overlapSketchThreshold 6, m_k 20, m_w 50, max_chain_iter 400, edge_threshold 4e6
        nsgpu_ctx *ctx;
        ns_check(nsgpu_create(&p, &ctx));
        uint32_t n_reads = 0;
        ns_check(nsgpu_load_fastq(ctx, text.data(), text.size(), &n_reads));
        std::cout << "numReads " << n_reads << "\n";

        std::vector<uint64_t> salts(p.n);                  // MinHashReadFilter::generateRandomNumbers (seed fixed here for a reproducible test)
        { std::mt19937_64 gen(12345); for (auto &x : salts) x = gen(); }
        ns_check(nsgpu_sketch(ctx, salts.data(), nullptr));
        ns_check(nsgpu_build_index(ctx));

        nsgpu_consensus_stats st;
        ns_check(nsgpu_consensus_run(ctx, /*n_builders=*/64, /*n_threads_out=*/numThr, &st));
        ns_check(nsgpu_consensus_write(ctx, tempDir.c_str(), tempFileName.c_str()));
        uint64_t bad = 0;
        ns_check(nsgpu_consensus_verify(ctx, &bad));
        std::cout << "numContigs = " << st.n_contigs << "\n#LoneReads = " << st.n_lone << "\nMinHash passed " << st.count_minhash
                  << " reads\nMinHash passed & not already in graph " << st.count_minhash_not_in_graph << " reads\nAligner passed " << st.count_aligner
                  << " reads\nlossless check: " << bad << " bad reads\n";
        nsgpu_destroy(ctx);
        return bad == 0 ? 0 : 1;
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        return 3;
    }
}
