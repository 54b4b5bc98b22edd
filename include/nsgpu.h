/*
 * nsgpu.h -- C-ABI of libnsgpu.so, the MI355X (gfx950) implementation of
 * NanoSpring's read-clustering + reference-encode hot path.
 *
 * Every entry point replaces one seam of the reference (paths relative to the
 * NanoSpring tree); INTEGRATION.md shows the C++ adaptor a maintainer adds on the
 * reference side.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns 0 on success and a negative nsgpu_status on error;
 *     nsgpu_last_error() returns a thread-local message (the adaptor rethrows it
 *     as std::runtime_error, as the reference does, src/main.cpp:161-176).
 *   - inputs are borrowed for the duration of the call; outputs named *_out are
 *     caller-allocated; outputs returned through T** are library-allocated and
 *     released with nsgpu_free().
 *   - every call is host-synchronous: results are complete (and host-visible) when it returns.
 *     The read store, the MinHash stages and the direct batch calls issue their device work on the
 *     context's stream (nsgpu_set_stream; default: a stream the context creates).  The contig stage,
 *     the batched minimizer sketches and the alignment DP additionally use PRIVATE non-blocking
 *     streams of the context (one per pipeline role and DP batch in flight, plus highest-priority side
 *     streams for the long DP problems); they are ordered against each other inside the library and
 *     drained before the call returns.  A caller that shares the GPU with other work on its own
 *     streams therefore needs no extra synchronisation, but cannot order INTO the middle of a call.
 *   - base alphabet is the reference's: code(c) = (c & 2) | ((c & 4) >> 2), i.e.
 *     A0 T1 C2 G3, N (and anything else) folds through the same bits
 *     (src/dnaToBits.cpp:6-8); 2-bit packing is MSB-first, 4 bases per byte
 *     (src/dnaToBits.cpp:10-36).
 */
#ifndef NSGPU_H_
#define NSGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nsgpu_ctx nsgpu_ctx;

typedef enum {
    NSGPU_OK = 0,
    NSGPU_ERR_ARG = -1,      /* bad argument / call order */
    NSGPU_ERR_HIP = -2,      /* HIP runtime error (message has hipGetErrorString) */
    NSGPU_ERR_NOMEM = -3,
    NSGPU_ERR_RANGE = -4,    /* value exceeds a format limit (e.g. > 2^32-1 reads, "Too many reads for read_t", src/ReadData.cpp:194-196) */
    NSGPU_ERR_NODEV = -5     /* no gfx950 device visible: the product path never falls back to the CPU */
} nsgpu_status;

/* The public fields of Compressor / MinHashReadFilter / Consensus that reach the
 * hot path (include/Compressor.h:10-35, include/ReadFilter.h:36-44,
 * include/Consensus.h:37-63). */
typedef struct {
    uint32_t k;                  /* MinHash k-mer size, -k (default 23); 1 <= k <= 31 */
    uint32_t n;                  /* sketch size, -n / --num-hash (default 60); 1 <= n <= 256 */
    uint32_t overlap_sketch_thr; /* --overlap-sketch-thr (default 6) */
    uint32_t m_k;                /* --minimap-k (default 20), <= 28 */
    uint32_t m_w;                /* --minimap-w (default 50), < 256 */
    uint32_t max_chain_iter;     /* --max-chain-iter (default 400) */
    uint64_t edge_threshold;     /* --edge-thr (default 4000000) */
    int32_t  device;             /* HIP device ordinal */
    int32_t  reserved;
} nsgpu_params;

void nsgpu_default_params(nsgpu_params *p);

/* ---- life cycle ---------------------------------------------------------- */
int  nsgpu_create(const nsgpu_params *p, nsgpu_ctx **ctx_out);
void nsgpu_destroy(nsgpu_ctx *ctx);
void nsgpu_free(void *p);
const char *nsgpu_last_error(void);
/* hipStream_t as void*; NULL restores the context's own stream. */
int  nsgpu_set_stream(nsgpu_ctx *ctx, void *hip_stream);
int  nsgpu_sync(nsgpu_ctx *ctx);
/* library build id + "gfx950"; never NULL */
const char *nsgpu_version(void);

/* ---- read store: replaces ReadData::loadFromFastqFile_lowmem's DnaBitset temp
 *      file + ReadData::getRead (src/ReadData.cpp:156-235) ----------------------
 * Reads become 2-bit packed rows resident in HBM (each row 16-byte aligned). */
/* ASCII in: bases = concatenation of N reads, off[N+1] byte offsets. Packing runs on the GPU. */
int nsgpu_load_reads_ascii(nsgpu_ctx *ctx, const char *bases, const uint64_t *off, uint32_t n_reads);
/* Already-packed in (the bytes DnaBitset::to_file writes, src/dnaToBits.cpp:100-103):
 * read r occupies packed[byte_off[r] .. byte_off[r] + (len[r]+3)/4). */
int nsgpu_load_reads_packed(nsgpu_ctx *ctx, const uint8_t *packed, const uint64_t *byte_off,
                            const uint32_t *len, uint32_t n_reads);
/* ---- f3 (the step in front of the path): ReadData::loadFromFastqFile (src/ReadData.cpp:78-221) for plain FASTQ text
 *      already in host memory (gzip is the caller's business).  Parsed on the GPU with std::getline's rules: read r =
 *      the whole line 4r+1 (a '\r' counts as a base and folds through baseToInt like every other byte), a missing base
 *      line is a read of length 0, an unterminated last line counts when non-empty.  At most 4 GiB of text per call.
 *      Errors like the reference: no reads, or 2^32-1 reads ("Too many reads for read_t type to handle."). ---- */
int nsgpu_load_fastq(nsgpu_ctx *ctx, const char *text, size_t n_bytes, uint32_t *n_reads_out);
/* The same for inputs of any size, piece by piece (the reference's logs are 85-130 Gbases): consecutive pieces of the file in order,
 * cut anywhere -- a record may straddle pieces; each piece < 3.75 GiB.  The result (rows in HBM, read ids, nsgpu_get_read) is what
 * ONE call over the whole text gives. */
int nsgpu_load_fastq_begin(nsgpu_ctx *ctx);
int nsgpu_load_fastq_chunk(nsgpu_ctx *ctx, const char *text, size_t n_bytes);
int nsgpu_load_fastq_end(nsgpu_ctx *ctx, uint32_t *n_reads_out);
/* ReadData::loadFromFile(fileName, FASTQ, gzip_flag) (src/ReadData.cpp:12-26; gzip through boost::iostreams::gzip_decompressor,
 * :95-101, :165-171): the file itself.  gzip_flag 0 = plain text, 1 = gzip (concatenated members are read through, like zcat),
 * -1 = decide by the file's first two bytes.  zlib inflates piece by piece on the host while the GPU parses the piece before; the
 * result is what nsgpu_load_fastq over the decompressed text gives.  "Can't open input file" like the reference (:79-81). */
int nsgpu_load_fastq_file(nsgpu_ctx *ctx, const char *path, int gzip_flag, uint32_t *n_reads_out);
uint32_t nsgpu_num_reads(const nsgpu_ctx *ctx);
uint64_t nsgpu_num_bases(const nsgpu_ctx *ctx);
/* ReadData::getRead (src/ReadData.cpp:225-235): read r as ASCII (A/T/C/G), out must hold len[r] bytes. */
int nsgpu_get_read(nsgpu_ctx *ctx, uint32_t r, char *out, uint32_t *len_out);
/* DnaBitset bytes of read r (checker for a1); out must hold (len+3)/4 bytes. */
int nsgpu_get_read_packed(nsgpu_ctx *ctx, uint32_t r, uint8_t *out, uint32_t *len_out);

/* ---- MinHash sketch + bucket tables: replaces MinHashReadFilter::initialize
 *      (src/ReadFilter.cpp:11-47).  The n salts are an explicit input (the
 *      reference draws them from std::random_device, src/ReadFilter.cpp:49-63). */
/* string2Sketch for every read (src/ReadFilter.cpp:117-136). sketches_out: N*n u64 row-major by read, or NULL. */
int nsgpu_sketch(nsgpu_ctx *ctx, const uint64_t *salts, uint64_t *sketches_out);
/* Multi-GPU: sketch only reads lo..hi (this rank's id range); rows of the sketch table can be read out / written
 * in place (host or device buffers, e.g. the buffers of an RCCL all-gather); nsgpu_sketch_mark_complete declares the
 * table complete once every row has been sketched or imported. */
int nsgpu_sketch_range(nsgpu_ctx *ctx, const uint64_t *salts, uint32_t lo, uint32_t hi);
int nsgpu_sketch_rows_get(nsgpu_ctx *ctx, uint32_t lo, uint32_t hi, void *dst, int dst_on_device);
int nsgpu_sketch_rows_set(nsgpu_ctx *ctx, uint32_t lo, uint32_t hi, const void *src, int src_on_device);
int nsgpu_sketch_mark_complete(nsgpu_ctx *ctx);
/* populateHashTables (src/ReadFilter.cpp:159-172; BBHashMap::initialize, src/BBHashMap.cpp:10-99). */
int nsgpu_build_index(nsgpu_ctx *ctx);
/* Table j as (distinct keys ascending, CSR start, ascending read ids) -- checker for a7.
 * keys_out: N u64, start_out: N+1 u32, ids_out: N u32; *nkeys_out distinct keys. */
int nsgpu_index_export(nsgpu_ctx *ctx, uint32_t j, uint64_t *keys_out, uint32_t *start_out,
                       uint32_t *ids_out, uint32_t *nkeys_out);

/* ---- overlap candidates: replaces ReadFilter::getFilteredReads
 *      (include/ReadFilter.h:24, src/ReadFilter.cpp:65-97) ------------------- */
/* One query string; *ids is library-allocated (nsgpu_free), ascending read ids. */
int nsgpu_filter(nsgpu_ctx *ctx, const char *s, size_t len, uint32_t **ids, size_t *n_ids);
/* Q query strings (strs + qoff[Q+1]); results as CSR: (*out_off)[Q+1], (*out_ids)[(*out_off)[Q]]. */
int nsgpu_filter_batch(nsgpu_ctx *ctx, const char *strs, const uint64_t *qoff, uint32_t n_queries,
                       uint64_t **out_off, uint32_t **out_ids);
/* Whole-read queries for every loaded read, forward (query 2r) and reverse-complement
 * (query 2r+1) -- the first window of every contig in Consensus::addRelatedReads
 * (src/Consensus.cpp:170-189).  Device resident; returns the total candidate count. */
int nsgpu_filter_all_reads(nsgpu_ctx *ctx, uint64_t *n_candidates_out);
/* copy the result of nsgpu_filter_all_reads to the host: off_out 2N+1 u64, ids_out n_candidates u32 */
int nsgpu_filter_all_fetch(nsgpu_ctx *ctx, uint64_t *off_out, uint32_t *ids_out);

/* ---- Consensus::checkRepetitive for every read (src/Consensus.cpp:405-442) -- */
int nsgpu_check_repetitive(nsgpu_ctx *ctx, uint8_t *flags_out);

/* ---- a14h: batched ksw_extd2 (minimap2/ksw2_extd2_sse.c:34-401 behind mm_align_pair,
 *      minimap2/align.c:313-339).  Problem i aligns query seqs[qoff[i] .. +qlen[i]) to target
 *      seqs[toff[i] .. +tlen[i]); bases are minimap2's codes 0..3, 4 = ambiguous
 *      (minimap2/sketch.c:9-26).  Scoring is ksw_gen_simple_mat(a, b, sc_ambi)
 *      (align.c:9-22) with gap costs (q,e)/(q2,e2).  flag takes the KSW_EZ_* bits of
 *      minimap2/ksw2.h:8-18 except KSW_EZ_GENERIC_SC / splice bits (unreachable from
 *      NanoSpring).  ez_out[i] = ksw_extz_t fields; CIGARs come back as a CSR. ---- */
typedef struct { int32_t a, b, sc_ambi, q, e, q2, e2; } nsgpu_ksw_params;
typedef struct {
    uint32_t max; int32_t zdropped;
    int32_t max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
} nsgpu_ksw_ez;
int nsgpu_ksw_extd2_batch(nsgpu_ctx *ctx, uint32_t n, const uint8_t *seqs, const uint64_t *qoff, const int32_t *qlen,
                          const uint64_t *toff, const int32_t *tlen, const int32_t *w, const int32_t *zdrop,
                          const int32_t *end_bonus, const int32_t *flag, const nsgpu_ksw_params *prm, nsgpu_ksw_ez *ez_out,
                          uint64_t **cigar_off_out, uint32_t **cigar_out);

/* ---- a14c: mm_sketch (minimap2/sketch.c:77-143; called by mm_idx_str -> mm_idx_add and by mm_map's
 *      collect_minimizers, minimap2/map.c:55-67) for a batch of sequences, rid = 0: the (w,k)-minimizers of
 *      sequence i (seqs[seq_off[i]..seq_off[i+1]), ASCII) are the mm128_t pairs
 *      (*xy_out)[2*j], (*xy_out)[2*j+1] = x, y for j in (*off_out)[i] .. (*off_out)[i+1], in the
 *      reference's output order (x = hash<<8 | span, y = pos<<1 | strand).  Both arrays are
 *      malloc'ed; release with nsgpu_free. ---- */
int nsgpu_mm_sketch_batch(nsgpu_ctx *ctx, const char *seqs, const uint64_t *seq_off, uint32_t n, uint32_t w, uint32_t k,
                          uint64_t **xy_out, uint64_t **off_out);

/* ---- a14e: the forward pass of mm_chain_dp (minimap2/chain.c:43-92; called by mm_map_frag, minimap2/map.c:295) for a
 *      batch of anchor lists: list i is the mm128_t pairs xy[2*j], xy[2*j+1] = x (rev<<63 | rid<<32 | ref pos),
 *      y (query span<<32 | query pos) for j in off[i] .. off[i+1], sorted by x as mm_map_frag leaves them.
 *      f_out[j] / p_out[j] are chain.c's f[] (best chain score ending in anchor j) and p[] (its predecessor, as an
 *      index into the same list, or -1), computed with minimap2's default max_gap = 5000, bw = 500,
 *      max_chain_skip = 25, chain_gap_scale = 1 and params.max_chain_iter.  nsgpu_align_batch and the contig
 *      engine run the same kernel; this entry exists so that it can be checked on its own. ---- */
int nsgpu_chain_scores(nsgpu_ctx *ctx, const uint64_t *xy, const uint64_t *off, uint32_t n, int32_t *f_out, int32_t *p_out);

/* ---- a14a/b/d: what mm_idx_str + mm_idx_cal_max_occ (minimap2/index.c:164-248, 365-402) and collect_seed_hits
 *      (minimap2/map.c:215-247, with MM_F_FOR_ONLY: map.c:139-145) leave for the chaining, for a batch of pairs:
 *      reference list r / query list i are mm128_t pairs as nsgpu_mm_sketch_batch returns them, pair i is
 *      (pair_ref[i], query i).  (*xy_out)[2*j], [2*j+1] for j in (*off_out)[i] .. (*off_out)[i+1] are pair i's
 *      anchors sorted by reference position (x = ref pos, y = span<<32 | query pos, bit 42 = tandem), mid_occ_out[i]
 *      the occurrence cut-off of its reference (mid_occ_frac 2e-4), avg_out[i] the mean query span of its anchors.
 *      flags_out[i] != 0: the kernel did not decide the pair (bit 0: two anchors on one reference position, whose
 *      order is the reference radix sort's; bit 1: more anchors than the LDS sort takes; bit 2: output capacity;
 *      bit 3: occurrence count above the histogram; bit 4: reference coordinate above 2^32) and returns no anchors
 *      for it -- nsgpu_align_batch and the contig engine redo such pairs with the literal host code.
 *      The arrays behind xy_out / off_out are malloc'ed; release with nsgpu_free. ---- */
int nsgpu_seed_anchors(nsgpu_ctx *ctx, const uint64_t *ref_xy, const uint64_t *ref_off, uint32_t n_refs, const uint64_t *qry_xy,
                       const uint64_t *qry_off, const uint32_t *pair_ref, uint32_t n_pairs, uint64_t **xy_out, uint64_t **off_out,
                       int32_t *mid_occ_out, uint32_t *flags_out, float *avg_out);

/* ---- a13: batched ConsensusGraph::alignRead (include/ConsensusGraph.h:245-247,
 *      src/ConsensusGraph.cpp:161-398): align query i (qrys[qry_off[i]..qry_off[i+1])) to reference
 *      pair_ref[i] (refs[ref_off[r]..ref_off[r+1])) with minimap2's defaults + MM_F_CIGAR|MM_F_FOR_ONLY,
 *      k = params.m_k, w = params.m_w, max_chain_iter = params.max_chain_iter, and convert reg[0] to
 *      the reference's Edit list.  Many queries may share one reference string (its minimizer index
 *      is built once per batch).  out[i].ok is alignRead's bool; hits == 0 means mm_map found nothing.
 *      Edit types: 0 SAME(num) 1 INSERT(base) 2 DELETE(base)  (include/Edits.h:8). ---- */
typedef struct { uint8_t type; uint8_t base; uint16_t reserved; uint32_t num; } nsgpu_edit;
typedef struct {
    int32_t ok, hits;
    int64_t rel_pos, begin_offset, end_offset;          /* relPos, beginOffset, endOffset */
    int32_t rs, re, qs, qe, blen, mlen, n_ambi, dp_max; /* mm_reg1_t / mm_extra_t fields of reg[0] */
    uint32_t n_cigar, n_edits;
    uint64_t cigar_off, edit_off;                       /* into *cigars_out / *edits_out */
} nsgpu_aln;
int nsgpu_align_batch(nsgpu_ctx *ctx, const char *refs, const uint64_t *ref_off, uint32_t n_refs, const char *qrys,
                      const uint64_t *qry_off, const uint32_t *pair_ref, uint32_t n_pairs, nsgpu_aln *out,
                      uint32_t **cigars_out, nsgpu_edit **edits_out);
/* cumulative counters of the align path since nsgpu_reset_align_stats (for bench.py) */
typedef struct {
    uint64_t pairs, dp_tasks, dp_rounds;
    double dp_cells;          /* sum of qlen*tlen over all DP problems */
    double index_ms, host_ms, dp_ms, dp_kernel_ms;   /* host wall / host wall / wall around the DP batches / HIP-event wall of the launches of a batch */
    double dp_kernel_sum_ms;  /* sum of the individual ksw_extd2 kernel durations (launches overlap on side streams) */
    double dp_alg_bytes;      /* sum over DP problems of qlen + tlen + 4 * n_cigar + sizeof(result) */
    uint64_t dp_launches;     /* ksw_extd2 kernel launches (one per LDS size class per DP round) */
    uint32_t host_threads, reserved;
    uint64_t seed_pairs_gpu;  /* pairs whose index + seeds + chaining scores came from the GPU kernels (seeds.hip, chain.hip) */
    uint64_t seed_pairs_host; /* pairs the seeding kernel handed back to the host code (anchors sharing a reference position, oversize lists) */
    /* the alignment plan on the device (plan.hip: chains, region, DP windows and the DP launch behind the chaining kernel without a host round
     * trip; the host's own plan looks the results up by key): */
    uint64_t plan_pairs_dev;  /* alignments whose DP problems the device planned and launched */
    uint64_t plan_pairs_host; /* alignments the plan kernel left to the host (several chains, seed filtering, span, capacity, no device input) */
    uint64_t plan_hits;       /* DP problems the host's plan asked for and found among the device-planned results */
    uint64_t plan_misses;     /* ... asked for and did not find (of alignments the device planned): launched in a later round */
    uint64_t plan_extra;      /* device-planned problems the host's plan did not ask for */
} nsgpu_align_stats;
int nsgpu_get_align_stats(const nsgpu_ctx *ctx, nsgpu_align_stats *s);
/* host waits for GPU work (stream / event waits inside the library) of this process so far.  No reference counterpart: the reference has no
 * device; with nsgpu_consensus_stats.n_rounds it gives the host <-> GPU hand-overs per slot of the contig stage (src/Consensus.cpp:29-137 is
 * one thread's loop there). */
uint64_t nsgpu_host_wait_count(void);
int nsgpu_reset_align_stats(nsgpu_ctx *ctx);

/* ---- a11/a12/a16/a17: the contig stage, replaces Consensus::generateAndWriteConsensus
 *      (src/Consensus.cpp:21-166) and everything below it.  Needs loaded reads, nsgpu_sketch and
 *      nsgpu_build_index.  n_builders virtual contig builders advance in lock-step rounds (all their
 *      window queries and alignments of a round are batched on the GPU); n_builders = 1 is the
 *      reference's deterministic `-t 1` schedule.  The builders' outputs are merged into
 *      n_threads_out stream sets, i.e. exactly the files Compressor::compress expects for numThr =
 *      n_threads_out (src/Compressor.cpp:111-143) plus metaData (src/Consensus.cpp:370-386). ---- */
typedef struct {
    uint32_t n_builders, reserved;
    uint64_t n_rounds, n_filter_rounds, n_align_rounds, n_windows, n_contigs, n_lone;
    uint64_t count_minhash, count_minhash_not_in_graph, count_aligner, n_align_calls;   /* CountStats, include/Consensus.h:19-35 */
    double total_ms, graph_ms, filter_ms, index_ms, align_ms;   /* wall per phase */
    double graph_cpu_ms, graph_max_ms;                          /* summed / longest single builder step in the graph phase */
    double graph_crit_ms, write_cpu_ms;                         /* sum over phases of the slowest builder step / CPU time in contig finalisation */
} nsgpu_consensus_stats;
/* Multi-GPU shards: global id of this context's read 0; the .id streams then carry global read ids so that the
 * stream sets of all shards can sit side by side as additional "threads" of one archive (default 0). */
int nsgpu_set_read_id_base(nsgpu_ctx *ctx, uint32_t base);
/* Schedule of the contig stage (all of nsgpu_consensus_run, nsgpu_dist_consensus_run and the phase calls below).
 *   groups            1, 2 or 4 (default): a builder steps once per `groups` slots.  4 = the pipelined engine for thousands of builders;
 *                     1 / 2 = few builders, where a slot is as long as its GPU round trips and the contig with the most reads sets the
 *                     run time.
 *   seed_bucket_depth 0 (default) = the reference's rule: the lowest unclaimed read at or after the builder's cursor
 *                     (Consensus::getRead, src/Consensus.cpp:444-468).  d >= 1 = conflict-aware seeds: reads are grouped into buckets of
 *                     the whole-read filter graph (every read without a bucket, in id order, takes the bucketless reads within d hops);
 *                     a contig in flight occupies the buckets of its reads; a new seed is the lowest unclaimed read of a bucket that
 *                     is neither occupied nor within seed_rings adjacency steps of an occupied one, builders in global order; a builder
 *                     that finds none waits.  Many concurrent builders then do not start contigs in each other's way (the reference's
 *                     -t N on a genome that is small for N threads fragments the same way; its streams grow with every extra contig).
 * With one builder every setting is the reference's -t 1 schedule.  Deterministic for fixed (reads, salts, builders, schedule). */
int nsgpu_set_schedule(nsgpu_ctx *ctx, uint32_t groups, uint32_t seed_bucket_depth, uint32_t seed_rings);
/* the same with a smaller exclusion radius (seed_tail_rings <= seed_rings) for the seed rounds in which more than half of ALL builders
 * ask for a seed -- the tail of a run, when a few contigs close the last gaps and everybody else waits: a seed there is one more contig
 * and halves what is left of the gap.  A seed round carries the requests of ONE builder group, so with 2 or 4 groups (at most a half / a
 * quarter of all builders per round) the smaller radius never applies: the option acts with groups = 1.  nsgpu_set_schedule =
 * seed_tail_rings equal to seed_rings. */
int nsgpu_set_schedule2(nsgpu_ctx *ctx, uint32_t groups, uint32_t seed_bucket_depth, uint32_t seed_rings, uint32_t seed_tail_rings);
/* The schedule derived from the input -- what a drop-in caller wants: the reference has ONE knob, -t (src/main.cpp:46-78), and no way to know
 * what this library's five should be.  After nsgpu_build_index the library knows the read count, the bases and -- from the whole-read filter
 * results per read -- the coverage, hence the genome size; from those it derives one group, the seed rule's bucket depth and radii (deep
 * coverage of a small genome: small buckets; shallow coverage of a large one: depth 3, 5 rings) and, when nsgpu_consensus_run /
 * nsgpu_dist_consensus_run is called with 0 builders, the builder count (1 per 10 Mbases, within what the seed rule can keep busy): streams
 * within 5 % of the reference's own -t 8 on the inputs it was measured on.  nsgpu_consensus_run(ctx, 0, ...) on a context
 * whose schedule was never set does the same without this call.  nsgpu_get_schedule2 reports what a run used (all out-pointers optional).
 * Deterministic for fixed (reads, salts); independent of the rank count. */
int nsgpu_set_schedule_auto(nsgpu_ctx *ctx);
/* Deferred alignments (one-group schedules).  In lock step a slot lasts as long as its slowest alignment, and a read across a tandem repeat --
 * 10^4 .. 10^5 anchors: milliseconds of chaining, then DP problems thousands of columns wide -- takes two to three times a whole slot.  With
 * slots >= 1 an alignment whose anchor list (collect_seed_hits's, minimap2/map.c:215-247, before chaining) is longer than `anchors` takes
 * `slots` MORE slots than the others: the builder stands still, the other builders' slots go on, and its result is delivered -- its claim
 * made -- at the end of slot s + slots.  The alignment itself is unchanged (the consensus it refers to does not move while it waits); the
 * schedule, as always, is a function of the data only, and the lock-step oracle states the same rule (oracle/consensus_oracle.cpp
 * LockStep::VT::extra).  The automatic schedule sets 4096 anchors / 2 slots unless this was called; explicit schedules leave it off.
 * nsgpu_get_defer: the setting and how many alignments the last contig stage deferred (out-pointers optional). */
int nsgpu_set_defer(nsgpu_ctx *ctx, uint32_t anchors, uint32_t slots);
int nsgpu_get_defer(const nsgpu_ctx *ctx, uint32_t *anchors, uint32_t *slots, uint64_t *n_deferred);
/* Where the contigs' consensus graphs live (ConsensusGraph: updateGraph, calculateMainPathGreedy, removeCycles / splitPath,
 * src/ConsensusGraph.cpp:400-807 -- the object behind Consensus::generateAndWriteConsensus's `cG`, src/Consensus.cpp:319-331).
 *   NSGPU_GRAPH_HOST    the pointer graph on the host, updated by the pool's threads (one update = ~0.1 ms of one core);
 *   NSGPU_GRAPH_DEVICE  a structure of arrays with 32-bit ids in HBM, one workgroup per update (one launch per slot whose workgroups wait
 *                       for the accepted reads' scripts), the finished contig copied back once for the edit emission;
 *   NSGPU_GRAPH_AUTO    (default) in HBM when the process has at most 5 host threads (a rank of a shared node: NSGPU_THREADS / the CPU
 *                       quota divided by the local ranks), else on the host -- the measured cross-over, DESIGN.md section 6.
 * `| NSGPU_GRAPH_CHECK`: every update in HBM is also run on the host by the same code with a team of one and the arrays are compared entry
 * by entry (tests).  The environment's NSGPU_GRAPH = host | device | auto applies while this was never called.  Both placements give the
 * same streams, byte for byte.  nsgpu_get_graph_stats: what the last contig stage used and what the kernels did. */
enum { NSGPU_GRAPH_AUTO = 0, NSGPU_GRAPH_HOST = 1, NSGPU_GRAPH_DEVICE = 2, NSGPU_GRAPH_CHECK = 0x100 };
typedef struct {
    uint32_t placement, checked;        /* NSGPU_GRAPH_HOST or _DEVICE; every update compared with the host's arrays */
    uint64_t n_updates, n_launches;     /* accepted reads put into graphs in HBM; kernel launches that served them */
    uint64_t n_array_growths, n_long_reports;   /* arrays re-allocated; new consensus stretches too long for a report (copied by the host) */
    uint64_t n_sequential_updates, n_full_walks, n_split_calls;   /* updates whose excursions were taken one at a time; removeCycles by the reference's
                                                                     full walk (the list of noted nodes did not account for all); splitPath calls */
    double kernel_ms[8];                /* the kernels' own clock, summed over the updates: tables, runs along the path, excursions, choices,
                                           stitching, writing the path, flags, removeCycles */
    double report_ms;                   /* script handed over -> consensus reported, summed */
    double host_wait_first_ms, host_wait_second_ms;   /* host threads waiting for the first (consensus) / second (after removeCycles) report */
    uint64_t by_duration[8];            /* updates by kernel time: < 0.25 / 0.5 / 1 / 2 / 4 / 8 / 16 ms / more */
    double gb_copied_back, hbm_peak_gb, hbm_mapped_gb, pinned_peak_gb, pinned_mapped_gb;
} nsgpu_graph_stats;
int nsgpu_set_graph(nsgpu_ctx *ctx, uint32_t mode);
int nsgpu_get_graph_stats(const nsgpu_ctx *ctx, nsgpu_graph_stats *stats_out);
int nsgpu_get_schedule2(const nsgpu_ctx *ctx, uint32_t *groups, uint32_t *seed_bucket_depth, uint32_t *seed_rings, uint32_t *seed_tail_rings, uint32_t *builders);
int nsgpu_get_schedule(const nsgpu_ctx *ctx, uint32_t *groups, uint32_t *seed_bucket_depth, uint32_t *seed_rings);
int nsgpu_consensus_run(nsgpu_ctx *ctx, uint32_t n_builders, uint32_t n_threads_out, nsgpu_consensus_stats *stats_out);
/* stream `which` of output thread `thread`: 0 .genome 1 .lone 2 .id 3 .pos 4 .type 5 .base 6 .complement, 7 = metaData
 * (thread ignored).  *data_out is library-allocated (nsgpu_free). */
int nsgpu_consensus_stream(nsgpu_ctx *ctx, uint32_t thread, uint32_t which, uint8_t **data_out, size_t *len_out);
/* The same engine phase by phase, for multi-GPU jobs (one process per GPU, every rank holds all reads and the whole
 * bucket index; rank r owns the builders with gid % world == r).  Builders form G = nsgpu_cons_groups() pipeline groups
 * (group = (gid >> 3) % G, a function of the global id only; G = 4); in slot s = 0, 1, 2, ... with h = s % G, b = (s + 1) % G:
 *   slot(s)                         concurrently: host phase of group h (graph updates up to the next window / alignment
 *                                   request), part 1 of the GPU batches of group (s + G - 1) % G (minimizer sketches,
 *                                   consensus indexes, seeds / chains, launch of the alignment DP), part 2 of group b (its
 *                                   window lookups; DP results, alignment skeletons, edit scripts); the DP kernels of
 *                                   group (s + 2) % G stay in flight
 *   claim_requests(b), seed_requests(h) -> [ONE all-gather of both lists] -> claim_resolve, then seed_resolve,
 *                                   then advance(only_fresh = 1, h) if a contig started
 * until claim_resolve / seed_resolve report that every builder is done.
 * The *_resolve calls take the request lists of ALL ranks (any order) and apply them to a replicated claim table in
 * global builder order, so every rank stays in step and the result does not depend on the number of ranks.
 * nsgpu_consensus_run is exactly this loop with world = 1.  group = -1 addresses all builders.  Lists returned
 * through T** are library-allocated.  G is the context's schedule (nsgpu_get_schedule; 4 or 2 here: the one-group schedule grants seeds
 * between the host phase and the batches of a slot and runs only inside nsgpu_consensus_run / nsgpu_dist_consensus_run). */
int nsgpu_cons_begin(nsgpu_ctx *ctx, uint32_t n_builders_total, uint32_t rank, uint32_t world);
uint32_t nsgpu_cons_groups(void);
int nsgpu_cons_slot(nsgpu_ctx *ctx, uint32_t slot);
int nsgpu_cons_advance(nsgpu_ctx *ctx, int only_fresh, int group);
int nsgpu_cons_seed_requests(nsgpu_ctx *ctx, int group, uint32_t **gids_out, uint32_t **cursors_out, uint32_t *n_out);
int nsgpu_cons_seed_resolve(nsgpu_ctx *ctx, const uint32_t *gids, const uint32_t *cursors, uint32_t n, uint32_t *n_started_out, uint32_t *all_done_out);
int nsgpu_cons_batches(nsgpu_ctx *ctx, int group);
int nsgpu_cons_claim_requests(nsgpu_ctx *ctx, int group, uint32_t **gids_out, uint32_t **reads_out, uint32_t *n_out);
int nsgpu_cons_claim_resolve(nsgpu_ctx *ctx, const uint32_t *gids, const uint32_t *reads, uint32_t n, uint32_t *all_done_out);
int nsgpu_cons_finish(nsgpu_ctx *ctx, uint32_t n_threads_out, nsgpu_consensus_stats *stats_out);
/* writes temp_dir + temp_file_name + ".tid.<t>" + ext for every thread/extension and temp_dir + "metaData",
 * the names Consensus / ConsensusGraphWriter use (temp_dir must end in '/'). */
int nsgpu_consensus_write(nsgpu_ctx *ctx, const char *temp_dir, const char *temp_file_name);
/* decodes the produced streams with the logic of Decompressor::generateRead (src/Decompressor.cpp:252-314)
 * and counts the reads that do not come back identical (0 = lossless). */
int nsgpu_consensus_verify(nsgpu_ctx *ctx, uint64_t *n_bad_out);


/* ---- multi-GPU: one process per GPU; the reference has no counterpart (it is one process with OpenMP threads,
 *      src/Consensus.cpp:29) -- this is what lets Compressor::compress() use the 8 GPUs of a node.  Reads shard by id.
 *      A communicator carries the collectives: RCCL over xGMI (bound at run time; rank 0 makes the id and the host program
 *      hands it to the other ranks by whatever channel it has), or host callbacks (gloo / MPI / TCP of the caller; used by
 *      the tests).  One communicator = one thread at a time. ------------------------------------------------------------ */
#define NSGPU_COMM_ID_BYTES 128
typedef struct nsgpu_comm nsgpu_comm;
typedef struct {
    void *user;
    /* every rank contributes bytes_per_rank bytes; recv receives world * bytes_per_rank in rank order (host memory); 0 = ok */
    int (*all_gather)(void *user, const void *send, void *recv, uint64_t bytes_per_rank);
    /* send_bytes[p] go to peer p, recv_bytes[p] come from peer p; blocks are contiguous in rank order (host memory); 0 = ok */
    int (*all_to_all)(void *user, const void *send, const uint64_t *send_bytes, void *recv, const uint64_t *recv_bytes);
} nsgpu_comm_callbacks;
int nsgpu_comm_unique_id(uint8_t *id_out /* NSGPU_COMM_ID_BYTES */);
int nsgpu_comm_init_rccl(nsgpu_ctx *ctx, const uint8_t *id, uint32_t rank, uint32_t world, nsgpu_comm **comm_out);
int nsgpu_comm_init_callbacks(nsgpu_ctx *ctx, const nsgpu_comm_callbacks *cb, uint32_t rank, uint32_t world, nsgpu_comm **comm_out);
void nsgpu_comm_destroy(nsgpu_comm *comm);
/* Rank r passes its shard (reads [lo_r, hi_r) of ONE read set, shards in rank order); afterwards every rank holds all reads
 * (all-gather; ReadData::getRead must answer for any id, src/ReadData.cpp:225-235).  *lo_out / *hi_out = this rank's id range. */
int nsgpu_dist_load_reads(nsgpu_ctx *ctx, nsgpu_comm *comm, const char *bases, const uint64_t *off, uint32_t n_local, uint32_t *lo_out, uint32_t *hi_out);
/* MinHashReadFilter::initialize (src/ReadFilter.cpp:11-47) over the ranks: each sketches its own id range, then the n bucket
 * tables are built
 *   NSGPU_DIST_REPLICATE  from all-gathered sketch rows, all n tables sorted on every rank;
 *   NSGPU_DIST_ALLTOALL   table j owned by rank j % world: all-to-all(v) of the (slot, key, id) tuples to the owners (RCCL
 *                         send/recv to all peers at once: every xGMI link busy), each owner sorts its tables over all reads,
 *                         the sorted tables are all-gathered (window queries stay local).
 * Both give the index nsgpu_build_index gives one process, bit for bit. */
#define NSGPU_DIST_REPLICATE 0
#define NSGPU_DIST_ALLTOALL 1
int nsgpu_dist_sketch_index(nsgpu_ctx *ctx, nsgpu_comm *comm, const uint64_t *salts, int mode);
/* Bytes this rank has received in all-gathers / all-to-alls through the communicator so far (load, every sketch_index, every contig
 * stage), and the host memory its copy of ALL reads occupies (2-bit rows + offset tables: about 0.25 B/base + 20 B/read). */
int nsgpu_comm_stats(const nsgpu_comm *comm, uint64_t *bytes_all_gather, uint64_t *bytes_all_to_all, uint64_t *host_bytes_reads);
/* Consensus::generateAndWriteConsensus over the ranks: the slot schedule documented above nsgpu_cons_begin with ONE small
 * all-gather (claim + seed request lists) per slot -- three in the one-group schedule: seed requests after the host phase, the slot's
 * (builder, candidate read) pairs before the batches (a builder whose read nobody else aligns updates its graph while the DP still runs),
 * claims after them; rank r owns the builders with gid % world == r and writes its contigs
 * (global read ids) into its own n_threads_out stream sets.  The result does not depend on the number of ranks. */
int nsgpu_dist_consensus_run(nsgpu_ctx *ctx, nsgpu_comm *comm, uint32_t n_builders_total, uint32_t n_threads_out, nsgpu_consensus_stats *stats_out);

/* ---- SURVEY 8 row f4, the part of the back end with a unique answer: the block sorter -------------------------------------------
 * Replaces libbsc's bsc_bwt_encode (/root/reference/libbsc/bwt/bwt.cpp:46-79) as bsc::BSC_compress calls it per 48 MB block of a stream
 * file (/root/reference/src/bsc.cpp:1045-1057 -> libbsc.cpp bsc_compress -> bsc_bwt_encode), reached from Compressor::compress()
 * (/root/reference/src/Compressor.cpp:111-143).  in / out: n bytes of HOST memory (out may equal in).  out = T[n-1] followed by
 * T[SA[k]-1] for the suffixes in ascending order without the row of suffix 0 (the end of the block sorts below every byte);
 * *primary_index = rank of suffix 0 + 1.  aux_rate (a power of two; 0 = none): aux[j] = rank of suffix j * aux_rate + 1 for
 * j = 0 .. (n-1)/aux_rate (aux[0] = the primary index): libsais_bwt_aux's I[], from which bsc_bwt_encode derives its `indexes`
 * (indexes[t] = aux[t + 1] - 1, num_indexes = (n-1)/aux_rate).  *gpu_ms: device time from the first kernel to the last (HIP events),
 * *rounds: prefix-doubling rounds.  QLFC / LZMA2 stay the caller's. */
int nsgpu_bwt_block(nsgpu_ctx *ctx, const uint8_t *in, uint64_t n, uint8_t *out, int32_t *primary_index, uint32_t aux_rate, int32_t *aux,
                    uint32_t *n_aux, double *gpu_ms, uint32_t *rounds);

/* ---- timing of the last call of each stage, in ms, measured with HIP events on
 *      the context's stream (for bench.py's roofline object) ------------------ */
typedef struct {
    float pack_ms, sketch_ms, index_ms, filter_ms, repetitive_ms;
    float sketch_kernel_ms;   /* the xor-min kernel alone */
    float filter_kernel_ms;
    uint64_t filter_matches;  /* sum of M over queries of the last filter call */
} nsgpu_timing;
int nsgpu_get_timing(const nsgpu_ctx *ctx, nsgpu_timing *t);

/* ---- synthetic reads (SURVEY 8d model; host side, deterministic per seed):
 *      iid genome of length G, reads start uniform, strand 50/50, length
 *      max(500, Gamma(2, mean/2)), per-base sub/ins/del at the given rates.
 *      Allocates *bases_out (ASCII, concatenated) and *off_out (n_reads+1). ---- */
int nsgpu_synth_reads(uint64_t seed, uint64_t genome_len, uint32_t n_reads, double mean_len,
                      double p_sub, double p_ins, double p_del,
                      char **bases_out, uint64_t **off_out);
/* reads [first, first + n_reads) of the same read set (multi-GPU ranks generate only their id range) */
int nsgpu_synth_reads_range(uint64_t seed, uint64_t genome_len, uint32_t first, uint32_t n_reads, double mean_len,
                            double p_sub, double p_ins, double p_del, char **bases_out, uint64_t **off_out);
/* the same read model over a genome with planted repeats (genome_kind 1: every ~40 kb in turn a 4 kb interspersed duplication, a 1.5 kb
 * tandem repeat, a homopolymer run, an (AT)n / (ACGT)n run); genome_kind 0 = the iid genome of the calls above */
int nsgpu_synth_reads_kind(uint64_t seed, uint64_t genome_len, uint32_t first, uint32_t n_reads, double mean_len, double p_sub,
                           double p_ins, double p_del, uint32_t genome_kind, char **bases_out, uint64_t **off_out);

#ifdef __cplusplus
}
#endif
#endif /* NSGPU_H_ */
