// host_util.hpp -- host thread pool helpers and the internal align-request interface.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>
#include "mm2.hpp"

struct nsgpu_ctx;

namespace nsgpu {

unsigned host_threads();
double now_ms();
void parallel_for_impl(size_t n, const std::function<void(size_t)> &fn);
template <class F> inline void par_for(size_t n, F fn) { parallel_for_impl(n, std::function<void(size_t)>(fn)); }
void parallel_for_pinned_impl(size_t n, const std::function<void(size_t)> &fn);
template <class F> inline void par_for_pinned(size_t n, F fn) { parallel_for_pinned_impl(n, std::function<void(size_t)>(fn)); }

struct AlignReq {
    const mm2::RefIndex *idx;     // index of `ref` (owned by the caller, reusable across batches)
    const char *ref; size_t ref_len;
    const char *qry; size_t qry_len;
};
int align_requests(nsgpu_ctx *c, std::vector<AlignReq> &reqs, std::vector<mm2::AlnOut> &outs);
int filter_strings_device(nsgpu_ctx *c, const char *strs, const uint64_t *qoff, uint32_t nq);   // api.hip: results stay in c->f_off / c->f_ids

}  // namespace nsgpu
