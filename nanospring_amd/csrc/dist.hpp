// dist.hpp -- the communicator the multi-GPU entry points talk to (dist.hip implements it over RCCL and over host callbacks).
#pragma once
#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime.h>

struct nsgpu_comm;

namespace nsgpu {

struct Comm {
    uint32_t rank = 0, world = 1;
    virtual ~Comm() {}
    // every rank contributes `bytes` bytes; recv holds world * bytes in rank order.  device: both pointers are device memory and the
    // call only ENQUEUES on st (RCCL) or returns after completion (callbacks); host buffers: complete on return
    virtual int all_gather(const void *send, void *recv, size_t bytes, bool device, hipStream_t st) = 0;
    // send_bytes[p] to peer p / recv_bytes[p] from peer p, blocks contiguous in rank order
    virtual int all_to_all_v(const void *send, const size_t *send_bytes, void *recv, const size_t *recv_bytes, bool device, hipStream_t st) = 0;
};

}  // namespace nsgpu

nsgpu::Comm *nsgpu_comm_impl(nsgpu_comm *c);
void nsgpu_comm_count(nsgpu_comm *c, uint64_t all_gather_bytes, uint64_t all_to_all_bytes);     // bench.py's record of collective traffic
