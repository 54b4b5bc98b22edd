// synth.cpp -- deterministic synthetic nanopore-like reads (host side, workload
// plumbing for bench.py and the tests; SURVEY 8d model, itself a restatement of
// the reference's util/old_code/createData.py length/error models):
//   genome  : iid uniform A/C/G/T of length G
//   read    : start uniform, strand 50/50, length max(500, Gamma(2, mean/2)),
//             per base: substitution / insertion / deletion at the given rates.
// Every read draws from its own counter-based generator, so the output does not
// depend on the number of threads.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <thread>
#include <vector>
#include <algorithm>
#include "../../include/nsgpu.h"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {   // splitmix64
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

inline uint64_t mix(uint64_t a, uint64_t b) { Rng r(a * 0xD6E8FEB86659FD93ull + b + 0x2545F4914F6CDD1Dull); r.next(); return r.next(); }

const char kBase[4] = {'A', 'C', 'G', 'T'};
inline char comp(char c) { switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; default: return 'C'; } }

struct Plan { uint64_t start; uint32_t len; bool rev; };

// emits the read (or just counts its length when out == nullptr)
uint64_t emit_read(const std::vector<char> &genome, const Plan &pl, uint64_t seed, uint32_t r, double p_sub, double p_ins, double p_del, char *out)
{
    Rng rng(mix(seed ^ 0xABCDEF12345ull, r));
    uint64_t o = 0;
    const double t1 = p_sub, t2 = p_sub + p_ins, t3 = p_sub + p_ins + p_del;
    for (uint32_t i = 0; i < pl.len; ++i) {
        char b = pl.rev ? comp(genome[pl.start + pl.len - 1 - i]) : genome[pl.start + i];
        const double u = rng.uni();
        if (u < t1) {
            char nb;
            do { nb = kBase[rng.next() & 3]; } while (nb == b);
            if (out) out[o]= nb;
            ++o;
        } else if (u < t2) {
            const char ib = kBase[rng.next() & 3];
            if (out) { out[o] = ib; out[o + 1] = b; }
            o += 2;
        } else if (u < t3) {
            /* deleted */
        } else {
            if (out) out[o] = b;
            ++o;
        }
    }
    return o;
}

}  // namespace

static int synth_impl(uint64_t seed, uint64_t genome_len, uint32_t first, uint32_t n_reads, double mean_len, double p_sub, double p_ins,
                      double p_del, char **bases_out, uint64_t **off_out, uint32_t genome_kind = 0);

// genome_kind 1: the iid genome with what real ONT genomes have and the iid one lacks, planted every ~150 kb in turn: a 4 kb segment copied
// from ~40 kb upstream (an interspersed duplication: minimizers with two reference positions, several chains, mm_set_parent / select_sub),
// a 1.5 kb tandem repeat of a 5-mer-to-40-mer unit (anchors sharing reference positions: the radix sort's tie order), a homopolymer run
// of 20-60 bases, and a 60-200 base (AT)n / (ACGT)n run (k-mers equal to their own reverse complement: mm_sketch pushes nothing there).
extern "C" int nsgpu_synth_reads_kind(uint64_t seed, uint64_t genome_len, uint32_t first, uint32_t n_reads, double mean_len, double p_sub,
                                      double p_ins, double p_del, uint32_t genome_kind, char **bases_out, uint64_t **off_out)
{
    return synth_impl(seed, genome_len, first, n_reads, mean_len, p_sub, p_ins, p_del, bases_out, off_out, genome_kind);
}

static void plant_repeats(std::vector<char> &g, uint64_t seed)
{
    const uint64_t L = g.size();
    uint64_t k = 0;
    for (uint64_t pos = 60000; pos + 8000 < L; pos += 37000 + (mix(seed ^ 0x77ull, k) % 3000), ++k) {
        Rng rng(mix(seed ^ 0x9999ull, k));
        switch (k & 3) {
        case 0: { const uint64_t src = pos - 40000 + rng.next() % 2000; for (uint64_t i = 0; i < 4000; ++i) g[pos + i] = g[src + i]; break; }
        case 1: { const uint64_t u = 5 + rng.next() % 36; for (uint64_t i = u; i < 1500; ++i) g[pos + i] = g[pos + i % u]; break; }
        case 2: { const uint64_t n = 20 + rng.next() % 41; const char b = kBase[rng.next() & 3]; for (uint64_t i = 0; i < n; ++i) g[pos + i] = b; break; }
        default: { const uint64_t n = 60 + rng.next() % 141; const char *unit = (rng.next() & 1) ? "AT" : "ACGT"; const uint64_t u = strlen(unit);
                   for (uint64_t i = 0; i < n; ++i) g[pos + i] = unit[i % u]; break; }
        }
    }
}

extern "C" int nsgpu_synth_reads(uint64_t seed, uint64_t genome_len, uint32_t n_reads, double mean_len, double p_sub, double p_ins,
                                 double p_del, char **bases_out, uint64_t **off_out)
{
    return synth_impl(seed, genome_len, 0, n_reads, mean_len, p_sub, p_ins, p_del, bases_out, off_out);
}

// reads [first, first + n_reads) of the read set that nsgpu_synth_reads(seed, genome_len, ...) defines
extern "C" int nsgpu_synth_reads_range(uint64_t seed, uint64_t genome_len, uint32_t first, uint32_t n_reads, double mean_len, double p_sub,
                                       double p_ins, double p_del, char **bases_out, uint64_t **off_out)
{
    return synth_impl(seed, genome_len, first, n_reads, mean_len, p_sub, p_ins, p_del, bases_out, off_out);
}

static int synth_impl(uint64_t seed, uint64_t genome_len, uint32_t first, uint32_t n_reads, double mean_len, double p_sub, double p_ins,
                      double p_del, char **bases_out, uint64_t **off_out, uint32_t genome_kind)
{
    if (!bases_out || !off_out || genome_len < 1000) return NSGPU_ERR_ARG;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 64) nt = 64;
    std::vector<char> genome(genome_len);
    {
        const uint64_t blk = 1 << 16, nblk = (genome_len + blk - 1) / blk;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
                for (uint64_t b = t; b < nblk; b += nt) {
                    Rng rng(mix(seed, b));
                    const uint64_t e = std::min(genome_len, (b + 1) * blk);
                    for (uint64_t i = b * blk; i < e; i += 32) {
                        uint64_t w = rng.next();
                        for (uint64_t j = i; j < std::min(e, i + 32); ++j, w >>= 2) genome[j] = kBase[w & 3];
                    }
                }
            });
        for (auto &x : th) x.join();
    }
    if (genome_kind == 1 && genome_len > 200000) plant_repeats(genome, seed);
    std::vector<Plan> plan(n_reads);
    for (uint32_t r = 0; r < n_reads; ++r) {
        Rng rng(mix(seed ^ 0x5151515151ull, (uint64_t)first + r));
        double l = -(mean_len / 2.0) * (std::log(1.0 - rng.uni()) + std::log(1.0 - rng.uni()));
        if (l < 500.0) l = 500.0;
        uint64_t len = (uint64_t)l;
        if (len > genome_len) len = genome_len;
        plan[r].len = (uint32_t)len;
        plan[r].start = (uint64_t)(rng.uni() * (double)(genome_len - len + 1));
        if (plan[r].start + len > genome_len) plan[r].start = genome_len - len;
        plan[r].rev = (rng.next() & 1) != 0;
    }
    uint64_t *off = (uint64_t *)malloc(((size_t)n_reads + 1) * sizeof(uint64_t));
    if (!off) return NSGPU_ERR_NOMEM;
    std::vector<uint64_t> lens(n_reads);
    auto run = [&](char *base) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
                for (uint32_t r = t; r < n_reads; r += nt) {
                    const uint64_t l = emit_read(genome, plan[r], seed, first + r, p_sub, p_ins, p_del, base ? base + off[r] : nullptr);
                    if (!base) lens[r] = l;
                }
            });
        for (auto &x : th) x.join();
    };
    run(nullptr);
    off[0] = 0;
    for (uint32_t r = 0; r < n_reads; ++r) off[r + 1] = off[r] + lens[r];
    char *bases = (char *)malloc(off[n_reads] + 16);
    if (!bases) { free(off); return NSGPU_ERR_NOMEM; }
    run(bases);
    *bases_out = bases;
    *off_out = off;
    return NSGPU_OK;
}
