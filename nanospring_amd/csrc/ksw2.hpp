// ksw2.hpp -- batch interface of the ksw_extd2 wavefront kernel (ksw2.hip).
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <hip/hip_runtime.h>

struct nsgpu_ctx;

namespace nsgpu {

struct KswTask {
    uint32_t qoff, toff;      // byte offsets of the 0..4 coded query / target in the sequence pool
    int32_t qlen, tlen;
    int32_t w, zdrop, end_bonus, flag;
    uint64_t p_off;           // traceback scratch offset (filled by ksw_run_batch)
    uint32_t cig_off, out_idx;
};

struct KswParams { int32_t sc_mch, sc_mis, sc_ambi, q, e, q2, e2; };   // sc_* as in the 5x5 matrix (mis, ambi negative)

struct KswResult {            // ksw_extz_t (minimap2/ksw2.h:23-32) without the pointer
    uint32_t max; int32_t zdropped;
    int32_t max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
};

size_t ksw_lds_bytes(int qlen, int tlen, int flag);
size_t ksw_p_bytes(int qlen, int tlen, int w);
int ksw_run_batch(nsgpu_ctx *c, std::vector<KswTask> &tasks, const uint8_t *seqs, size_t seq_bytes, const KswParams &pr,
                  std::vector<KswResult> &results, std::vector<uint32_t> &cigars, std::vector<uint64_t> &cig_off, int ws_index = 0);

int ksw_batch_launch(nsgpu_ctx *c, std::vector<KswTask> &tasks, const uint8_t *seqs, size_t seq_bytes, const KswParams &pr,
                     std::vector<KswResult> &results, std::vector<uint32_t> &cigars, std::vector<uint64_t> &cig_off, int ws_index);
int ksw_batch_collect(nsgpu_ctx *c, std::vector<KswTask> &tasks, std::vector<KswResult> &results, std::vector<uint32_t> &cigars,
                      std::vector<uint64_t> &cig_off, int ws_index);

// second generation (ksw2_reg.hip): DP state in registers, packed int16 arithmetic
#define KSW_REG_CLASSES 13
int ksw_reg_class(const KswTask &t, const KswParams &pr);          // 0 .. KSW_REG_CLASSES-1, or -1 (first-generation kernels)
int ksw_reg_cells(int cls);                                        // widest tlen the class serves
int ksw_reg_threads(int cls);
size_t ksw_reg_lds_bytes(int cls, int qlen);
struct DvCollect;          // ksw_collect.hpp: with it (device-planned batches) every problem counts itself on its alignment and the last one hands the alignment over
int ksw_reg_launch(int cls, hipStream_t st, uint32_t m, size_t lds_bytes, const KswTask *tasks, const uint32_t *order, const KswParams &pr, const uint8_t *seqs,
                   uint8_t *p_pool, uint32_t *cig_pool, KswResult *res, const uint32_t *n_dev = nullptr, const DvCollect *dc = nullptr);       // n_dev: the count lives in device memory, m bounds it

}  // namespace nsgpu
