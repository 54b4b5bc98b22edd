// The multi-GPU adaptor of INTEGRATION.md section 3b as a stand-alone C++11 program (no Python): W ranks -- here W processes forked before
// anything touches the GPU, all on device 0 -- each load their shard of ONE FASTQ file, join a communicator whose two collectives are
// host callbacks over a shared-memory segment (a C++ host would plug MPI here, or use nsgpu_comm_init_rccl), and run
//   nsgpu_dist_load_reads -> nsgpu_dist_sketch_index (all-to-all of bucket tuples) -> nsgpu_dist_consensus_run
// Every rank writes its stream sets; rank 0 prints the totals.  Built and run by tests/test_dist_gpu.py.
//   usage: dist_stage <reads.fastq> <tempDir/> <world> <builders> <groups> <seed_depth> <seed_rings>
#include "nsgpu.h"
#include <pthread.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

static void ns_check(int rc) { if (rc != NSGPU_OK) throw std::runtime_error(nsgpu_last_error()); }

// ---- the "MPI" of this test: a process-shared barrier and one staging area -------------------------------------------------------------
struct Shared {
    pthread_barrier_t bar;
    uint64_t counts[64 * 64];            // all-to-all: bytes rank r sends to rank p at [r * world + p]
    uint64_t results[64][8];
    unsigned char data[1];               // staging (size chosen at start)
};
struct Rank { Shared *sh; size_t cap; uint32_t rank, world; };

static int cb_all_gather(void *user, const void *send, void *recv, uint64_t bytes)
{
    Rank *R = static_cast<Rank *>(user);
    if (bytes * R->world > R->cap) return 1;
    memcpy(R->sh->data + (size_t)R->rank * bytes, send, bytes);
    pthread_barrier_wait(&R->sh->bar);
    memcpy(recv, R->sh->data, bytes * R->world);
    pthread_barrier_wait(&R->sh->bar);
    return 0;
}

static int cb_all_to_all(void *user, const void *send, const uint64_t *send_bytes, void *recv, const uint64_t *recv_bytes)
{
    Rank *R = static_cast<Rank *>(user);
    const uint32_t W = R->world;
    for (uint32_t p = 0; p < W; ++p) R->sh->counts[(size_t)R->rank * W + p] = send_bytes[p];
    pthread_barrier_wait(&R->sh->bar);
    // rank r's blocks start behind those of the ranks before it
    uint64_t my_base = 0, total = 0;
    for (uint32_t r = 0; r < W; ++r) for (uint32_t p = 0; p < W; ++p) { if (r < R->rank) my_base += R->sh->counts[(size_t)r * W + p]; total += R->sh->counts[(size_t)r * W + p]; }
    if (total > R->cap) return 1;
    uint64_t tot_send = 0;
    for (uint32_t p = 0; p < W; ++p) tot_send += send_bytes[p];
    memcpy(R->sh->data + my_base, send, tot_send);
    pthread_barrier_wait(&R->sh->bar);
    uint64_t ro = 0;
    for (uint32_t r = 0; r < W; ++r) {                      // block r -> me sits in r's area behind r's blocks to the ranks before me
        uint64_t base = 0;
        for (uint32_t q = 0; q < r; ++q) for (uint32_t p = 0; p < W; ++p) base += R->sh->counts[(size_t)q * W + p];
        for (uint32_t p = 0; p < R->rank; ++p) base += R->sh->counts[(size_t)r * W + p];
        const uint64_t n = R->sh->counts[(size_t)r * W + R->rank];
        if (n != recv_bytes[r]) return 2;
        memcpy(static_cast<unsigned char *>(recv) + ro, R->sh->data + base, n);
        ro += n;
    }
    pthread_barrier_wait(&R->sh->bar);
    return 0;
}

static int run_rank(Rank &R, const std::string &text_path, const std::string &tempDir, uint32_t builders, uint32_t groups, uint32_t depth, uint32_t rings)
{
    try {
        // the rank's shard: records [lo, hi) of the file (every rank reads the file; a real host would seek)
        std::ifstream in(text_path, std::ios::binary);
        std::string line;
        std::vector<std::string> reads;
        while (std::getline(in, line)) { std::getline(in, line); reads.push_back(line); std::getline(in, line); std::getline(in, line); }
        const uint32_t N = (uint32_t)reads.size(), lo = (uint32_t)((uint64_t)N * R.rank / R.world), hi = (uint32_t)((uint64_t)N * (R.rank + 1) / R.world);
        std::string bases;
        std::vector<uint64_t> off(1, 0);
        for (uint32_t r = lo; r < hi; ++r) { bases += reads[r]; off.push_back(bases.size()); }

        nsgpu_params p;
        nsgpu_default_params(&p);
        nsgpu_ctx *ctx;
        ns_check(nsgpu_create(&p, &ctx));
        ns_check(nsgpu_set_schedule(ctx, groups, depth, rings));
        nsgpu_comm_callbacks cb;
        cb.user = &R, cb.all_gather = cb_all_gather, cb.all_to_all = cb_all_to_all;
        nsgpu_comm *comm;
        ns_check(nsgpu_comm_init_callbacks(ctx, &cb, R.rank, R.world, &comm));
        uint32_t glo = 0, ghi = 0;
        ns_check(nsgpu_dist_load_reads(ctx, comm, bases.data(), off.data(), hi - lo, &glo, &ghi));
        if (glo != lo || ghi != hi) throw std::runtime_error("id range of the shard");
        std::vector<uint64_t> salts(p.n);
        { std::mt19937_64 gen(12345); for (auto &x : salts) x = gen(); }
        ns_check(nsgpu_dist_sketch_index(ctx, comm, salts.data(), NSGPU_DIST_ALLTOALL));
        nsgpu_consensus_stats st;
        ns_check(nsgpu_dist_consensus_run(ctx, comm, builders, 1, &st));
        const std::string name = "Stream.rank" + std::to_string(R.rank);
        ns_check(nsgpu_consensus_write(ctx, (tempDir + "rank" + std::to_string(R.rank) + "/").c_str(), "Stream"));
        uint64_t bad = 0, ag = 0, aa = 0, hb = 0;
        ns_check(nsgpu_consensus_verify(ctx, &bad));
        ns_check(nsgpu_comm_stats(comm, &ag, &aa, &hb));
        uint64_t *res = R.sh->results[R.rank];
        res[0] = st.n_contigs, res[1] = st.n_lone, res[2] = st.count_aligner, res[3] = bad, res[4] = ag, res[5] = aa, res[6] = hb, res[7] = nsgpu_num_bases(ctx);
        nsgpu_comm_destroy(comm);
        nsgpu_destroy(ctx);
        return bad == 0 ? 0 : 1;
    } catch (const std::exception &e) {
        std::cerr << "rank " << R.rank << " error: " << e.what() << "\n";
        return 3;
    }
}

int main(int argc, char **argv)
{
    if (argc < 8) { std::cerr << "usage: dist_stage <reads.fastq> <tempDir/> <world> <builders> <groups> <seed_depth> <seed_rings>\n"; return 2; }
    const uint32_t W = (uint32_t)std::stoul(argv[3]);
    if (W < 1 || W > 64) return 2;
    const size_t cap = 1ull << 30;
    Shared *sh = static_cast<Shared *>(mmap(nullptr, sizeof(Shared) + cap, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
    if (sh == MAP_FAILED) return 4;
    pthread_barrierattr_t ba;
    pthread_barrierattr_init(&ba);
    pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&sh->bar, &ba, W);
    std::vector<pid_t> kids;
    for (uint32_t r = 0; r < W; ++r) {
        const pid_t pid = fork();                        // before anything has touched the GPU
        if (pid == 0) {
            Rank R{sh, cap, r, W};
            _exit(run_rank(R, argv[1], argv[2], (uint32_t)std::stoul(argv[4]), (uint32_t)std::stoul(argv[5]), (uint32_t)std::stoul(argv[6]), (uint32_t)std::stoul(argv[7])));
        }
        kids.push_back(pid);
    }
    int worst = 0;
    for (pid_t k : kids) { int stt = 0; waitpid(k, &stt, 0); const int rc = WIFEXITED(stt) ? WEXITSTATUS(stt) : 9; if (rc > worst) worst = rc; }
    uint64_t tot[4] = {0, 0, 0, 0};
    for (uint32_t r = 0; r < W; ++r) {
        for (int i = 0; i < 4; ++i) tot[i] += sh->results[r][i];
        std::cout << "rank " << r << ": all-gather bytes " << sh->results[r][4] << " all-to-all bytes " << sh->results[r][5] << " host bytes of the read copy " << sh->results[r][6]
                  << " bases " << sh->results[r][7] << "\n";
    }
    std::cout << "numContigs = " << tot[0] << "\n#LoneReads = " << tot[1] << "\nAligner passed " << tot[2] << " reads\nlossless check: " << tot[3] << " bad reads\n";
    return worst;
}
