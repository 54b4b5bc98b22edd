// ksw_class.hpp -- which register-resident DP class serves a problem, and how much traceback scratch it needs: the ONE definition
// the host launch code (ksw2.hip / ksw2_reg.hip) and the device-side alignment plan (plan.hip) share, so that a problem planned on
// the device is launched exactly as the host would launch it.
#pragma once
#include <cstdint>
#include <cstddef>
#include <hip/hip_runtime.h>
#include "ksw2.hpp"

namespace nsgpu {

// the A/B switches of the class choice, read from the environment once on the host (ksw_class_config) and handed to device code by value
struct KswClassCfg {
    int32_t off;               // NSGPU_KSW_NO_REG: first-generation kernels only
    int32_t promote_rows;      // NSGPU_KSW_PROMOTE_ROWS (default 520; negative = off)
    int32_t flag_or;           // KSW_EZ_NS_* bits the host adds to every task
    int32_t long_rows;         // device-planned batches only: problems of the one-wave classes that can take more anti-diagonals than this run apart
                               // from the bulk (class 12: the <1,4> kernel on a side stream) -- what is left finishes early (NSGPU_KSW_LONG_ROWS; 0 = off)
};
const KswClassCfg &ksw_class_config();

__host__ __device__ inline size_t ksw_p_bytes_hd(int qlen, int tlen, int w)
{
    if (qlen <= 0 || tlen <= 0) return 0;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    return ((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16;
}

// cells per row the width classes hold: <1,2> <1,4> <12,1> (class 8) <8,5>.  Class numbers 2 and 4 .. 7 belonged to variants that were
// measured and lost (<4,3>; the latency twins <2,1> / <4,1>; <5,3> / <9,5> with a books wave: DESIGN.md section 4) and are retired.
__host__ __device__ inline int ksw_reg_width(int c) { return c == 0 ? 256 : c == 1 ? 512 : c == 2 ? 1536 : 5120; }

// ksw2_reg.hip's eligibility proofs + the class by target width; -1: not for the register kernels
__host__ __device__ inline int ksw_reg_class_hd(int qlen, int tlen, int w_in, int /*flag*/, const KswParams &pr, const KswClassCfg &cfg)
{
    if (cfg.off || qlen <= 0 || tlen <= 0) return -1;
    int q = pr.q, e = pr.e, q2 = pr.q2, e2 = pr.e2;
    if (q2 + e2 < q + e) { int t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int sc_n = pr.sc_ambi == 0 ? -e2 : pr.sc_ambi;
    if (pr.sc_mch < 0 || pr.sc_mch > 4 || pr.sc_mis > 0 || pr.sc_mis < -8 || sc_n > 0 || sc_n < -8) return -1;
    if (q < 0 || e < 1 || q + e > 12 || q2 + e2 > 32 || e2 < 1 || q2 < 0) return -1;
    int w = w_in;
    if (w < 0 || w > qlen + tlen) w = qlen + tlen;
    const int mn = qlen < tlen ? qlen : tlen;
    // |H| of any in-band cell stays a 16-bit key: H <= sc_mch * min(qlen, tlen); along a diagonal H drops by at most |sc_mis| per cell, a
    // cell entering the band starts at most q + e below its neighbour (u >= -(q + e) for sane states), and there are at most w + 1 diagonals
    if ((long long)(-pr.sc_mis > pr.sc_mch ? -pr.sc_mis : pr.sc_mch) * mn + (long long)(q + e) * (w + 1) + 64 >= 32768) return -1;
    for (int c = 0; c < 4; ++c)
        if (tlen <= ksw_reg_width(c)) {
            return c == 2 ? 8 : c;                                                              // (513 .. 1536 columns: <12,1>)
        }
    return -1;
}

// anti-diagonals the sweep of a problem can take: all of them, or until the band runs out (extensions that run off the target's end stop
// earlier still: the exact early exit, ksw2_reg.hip)
__host__ __device__ inline long long ksw_rows_bound(int qlen, int tlen, int w_in)
{
    const long long w = w_in < 0 ? (long long)qlen + tlen : w_in;
    const long long full = (long long)qlen + tlen - 1, band = 2ll * tlen + w + 1;
    return full < band ? full : band;
}

// the launch rule on top of it (ksw_batch_launch): the few LONG problems of the narrowest class go with the <1,4> launch.  two_phase (a
// device-planned batch whose results are fetched in two parts, ksw_dev_launch): the long problems of both one-wave classes form class 12
__host__ __device__ inline int ksw_launch_class_hd(int qlen, int tlen, int w_in, int flag, const KswParams &pr, const KswClassCfg &cfg, bool two_phase = false)
{
    int rcls = ksw_reg_class_hd(qlen, tlen, w_in, flag, pr, cfg);
    if (two_phase && cfg.long_rows > 0 && (rcls == 0 || rcls == 1) && ksw_rows_bound(qlen, tlen, w_in) > cfg.long_rows) return 12;
    if (rcls == 0 && cfg.promote_rows >= 0 && ksw_rows_bound(qlen, tlen, w_in) > cfg.promote_rows) rcls = 1;
    return rcls;
}
// a class whose problems a device-planned batch finishes late: everything that does not run on the main stream (ksw_dev_launch) -- the wide
// problems, the long ones, the variants behind switches
__host__ __device__ inline bool ksw_class_is_slow(int cls) { return cls >= 2; }

}  // namespace nsgpu
